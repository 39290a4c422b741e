/*
 * TEST INFRASTRUCTURE — CPU oracle (see pie_oracle.h for scope and the parity-pin statement).
 * Citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "pie_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------- synthetic corpus */

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define GOLDEN 0x9E3779B97F4A7C15ULL

/* n-th output (0-based) of the sequential splitmix64 generator seeded with `seed` */
static inline uint64_t sm_out(uint64_t seed, uint64_t n) { return mix64(seed + (n + 1) * GOLDEN); }

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

uint64_t pie_oracle_splitmix64(uint64_t x) { return mix64(x + GOLDEN); }

int32_t pie_oracle_shard_of(int32_t user, int32_t n_shards)
{
    return (int32_t)(pie_oracle_splitmix64((uint64_t)(uint32_t)user) % (uint64_t)n_shards);
}

void pie_oracle_gen(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                    uint32_t flags, int64_t *start, int64_t *end, int32_t *user, int32_t *disc)
{
    for (int64_t k = 0; k < n; ++k) {
        const uint64_t i = (uint64_t)(row0 + k);
        const uint64_t r0 = sm_out(seed, 4 * i + 0), r1 = sm_out(seed, 4 * i + 1);
        const uint64_t r2 = sm_out(seed, 4 * i + 2), r3 = sm_out(seed, 4 * i + 3);
        /* back-dated arithmetic timestamps, the pattern of scripts/simulate-archive.js:14-35 */
        int64_t s = PIE_ORACLE_T0_MS - (int64_t)mulhi64(r2, (uint64_t)PIE_ORACLE_SPAN_MS);
        if (flags & PIE_GEN_TIME_ORDERED) /* rows in order of creation: start ascending with the row index */
            s = PIE_ORACLE_T0_MS - PIE_ORACLE_SPAN_MS + 1 + (int64_t)(((unsigned __int128)i * (uint64_t)PIE_ORACLE_SPAN_MS) / (uint64_t)n_total);
        int64_t dur = PIE_ORACLE_TTL_MS; /* expiresAt = createdAt + SESSION_TTL_MS, server/sessionStore.js:15-16 */
        if (flags & PIE_GEN_INTERVAL)
            dur = PIE_ORACLE_MIN_DUR_MS + (int64_t)mulhi64(r3, (uint64_t)(PIE_ORACLE_TTL_MS - PIE_ORACLE_MIN_DUR_MS + 1));
        start[k] = s;
        end[k] = s + dur;
        user[k] = (flags & PIE_GEN_CLUSTERED) ? (int32_t)((i * (uint64_t)n_users) / (uint64_t)n_total)
                                              : (int32_t)mulhi64(r0, (uint64_t)n_users);
        disc[k] = (int32_t)mulhi64(r1, (uint64_t)n_disc);
    }
}

void pie_oracle_gen_cdf(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                        uint32_t flags, const uint64_t *cdf, int64_t *start, int64_t *end, int32_t *user, int32_t *disc)
{
    pie_oracle_gen(seed, n_total, row0, n, n_users, n_disc, flags & ~PIE_GEN_CLUSTERED, start, end, user, disc);
    for (int64_t k = 0; k < n; ++k) {
        const uint64_t r0 = sm_out(seed, 4 * (uint64_t)(row0 + k));
        int32_t lo = 0, hi = n_users - 1; /* first index with r0 < cdf[index]; the last threshold catches everything */
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (r0 < cdf[mid]) hi = mid; else lo = mid + 1;
        }
        user[k] = lo;
    }
}

/* ---------------------------------------------------------------- row predicate (SURVEY.md §8 a-D) */

int pie_oracle_selected(int64_t start, int64_t end, int32_t disc, int64_t now, int64_t cutoff, uint64_t disc_mask)
{
    /* dead iff expiresAt <= now: server/sessionStore.js:30 (getSession) and :69 (purgeExpiredSessions) */
    if (end <= now) return 0;
    /* window: Number.isFinite(startTs) && startTs >= cutoff, server/storage/sqlProvider.js:284.
     * int64 columns are finite by construction; non-finite inputs are rejected at the host boundary. */
    if (!(start >= cutoff)) return 0;
    /* discipline must resolve (findDiscipline !== null, server/disciplineConfig.js:88-97) and be enabled in
     * the predicate table; an id outside the table resolves to nothing. */
    if (disc < 0 || disc >= 64) return 0;
    if (!((disc_mask >> disc) & 1ULL)) return 0;
    return 1;
}

/* ---------------------------------------------------------------- per-bucket stable order */

typedef struct {
    int64_t start;
    int32_t idx;
} pair_t;

/* bottom-up stable merge sort by start only: equal starts keep their incoming (row-index) order.
 * ORDER BY start_ts ASC (server/storage/sqlProvider.js:276); tie rule = ascending row index (a-D), which is
 * Map insertion order (server/sessionStore.js:59,68) and what a stable JS sort gives (public/app.js:3004). */
static void stable_sort_pairs(pair_t *a, pair_t *tmp, size_t n)
{
    for (size_t i = 1; i < n; ++i) { /* insertion sort runs of 8 */
        if ((i & 7) == 0) continue;
        pair_t v = a[i];
        size_t j = i, lo = i & ~(size_t)7;
        while (j > lo && a[j - 1].start > v.start) { a[j] = a[j - 1]; --j; }
        a[j] = v;
    }
    pair_t *src = a, *dst = tmp;
    for (size_t w = 8; w < n; w <<= 1) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].start < src[i].start) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        pair_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(pair_t));
}

int pie_oracle_scan(const int64_t *start, const int64_t *end, const int32_t *user, const int32_t *disc, size_t n,
                    int32_t n_users, int64_t now, int64_t cutoff, uint64_t disc_mask, int32_t *counts,
                    int64_t *offsets, int32_t *idx, size_t idx_cap, size_t *m_out)
{
    memset(counts, 0, (size_t)n_users * sizeof(int32_t));
    /* one pass in row order == Map insertion order (server/sessionStore.js:59,68); `now` is one scalar for
     * the whole scan (sampled once, server/sessionStore.js:67) */
    size_t cap = 1024, m = 0;
    int32_t *sel = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!sel) return -3;
    for (size_t i = 0; i < n; ++i) {
        if (!pie_oracle_selected(start[i], end[i], disc[i], now, cutoff, disc_mask)) continue;
        const int32_t u = user[i];
        if (u < 0 || u >= n_users) { free(sel); return -2; }
        if (m == cap) {
            cap *= 2;
            int32_t *p = (int32_t *)realloc(sel, cap * sizeof(int32_t));
            if (!p) { free(sel); return -3; }
            sel = p;
        }
        sel[m++] = (int32_t)i;
        counts[u]++; /* user match: session.userId === userId, server/sessionStore.js:60 */
    }
    offsets[0] = 0;
    for (int32_t u = 0; u < n_users; ++u) offsets[u + 1] = offsets[u] + counts[u];
    if (m_out) *m_out = m;
    if (m > idx_cap) { free(sel); return -1; }
    if (m == 0) { free(sel); return 0; }

    pair_t *pairs = (pair_t *)malloc(2 * m * sizeof(pair_t));
    int64_t *cursor = (int64_t *)malloc((size_t)n_users * sizeof(int64_t));
    if (!pairs || !cursor) { free(sel); free(pairs); free(cursor); return -3; }
    memcpy(cursor, offsets, (size_t)n_users * sizeof(int64_t));
    for (size_t k = 0; k < m; ++k) { /* fill buckets in row order */
        const int32_t i = sel[k];
        pair_t *p = &pairs[cursor[user[i]]++];
        p->start = start[i];
        p->idx = i;
    }
    size_t maxc = 0;
    for (int32_t u = 0; u < n_users; ++u) if ((size_t)counts[u] > maxc) maxc = (size_t)counts[u];
    pair_t *tmp = pairs + m; /* second half is merge scratch (maxc <= m) */
    (void)maxc;
    for (int32_t u = 0; u < n_users; ++u)
        if (counts[u] > 1) stable_sort_pairs(pairs + offsets[u], tmp, (size_t)counts[u]);
    for (size_t k = 0; k < m; ++k) idx[k] = pairs[k].idx;
    free(sel); free(pairs); free(cursor);
    return 0;
}

/* ---------------------------------------------------------------- the same scan on several host threads (B2) */

typedef struct mt_job {
    /* inputs */
    const int64_t *start, *end;
    const int32_t *user, *disc;
    size_t n;
    int32_t n_users;
    int64_t now, cutoff;
    uint64_t mask;
    int n_threads, tid;
    /* shared state */
    struct mt_job *all;
    int32_t *counts;      /* global */
    int64_t *offsets;     /* global */
    pair_t *pairs, *tmp;  /* global, m entries each */
    int32_t *idx;
    pthread_barrier_t *bar;
    volatile int *err;
    size_t *m_shared;
    size_t idx_cap;
    /* per thread */
    int32_t *sel;         /* selected rows of this thread's chunk, in row order */
    size_t n_sel;
    int32_t *cnt;         /* per-user counts of this chunk, then this chunk's first slot per user */
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    const int T = j->n_threads, t = j->tid;
    const size_t lo = j->n * (size_t)t / (size_t)T, hi = j->n * (size_t)(t + 1) / (size_t)T;
    /* phase 1: select this chunk (row order == Map insertion order, server/sessionStore.js:59,68) */
    size_t cap = (hi - lo) / 64 + 1024, m = 0;
    int32_t *sel = (int32_t *)malloc(cap * sizeof(int32_t));
    int32_t *cnt = (int32_t *)calloc((size_t)j->n_users, sizeof(int32_t));
    if (!sel || !cnt) *j->err = -3;
    for (size_t i = lo; i < hi && !*j->err; ++i) {
        if (!pie_oracle_selected(j->start[i], j->end[i], j->disc[i], j->now, j->cutoff, j->mask)) continue;
        const int32_t u = j->user[i];
        if (u < 0 || u >= j->n_users) { *j->err = -2; break; }
        if (m == cap) {
            cap *= 2;
            int32_t *p = (int32_t *)realloc(sel, cap * sizeof(int32_t));
            if (!p) { *j->err = -3; break; }
            sel = p;
        }
        sel[m++] = (int32_t)i;
        cnt[u]++;
    }
    j->sel = sel; j->n_sel = m; j->cnt = cnt;
    pthread_barrier_wait(j->bar);
    if (*j->err) return NULL;
    /* phase 2: counts[u] = sum over chunks (users split over the threads) */
    const int32_t ulo = (int32_t)((int64_t)j->n_users * t / T), uhi = (int32_t)((int64_t)j->n_users * (t + 1) / T);
    for (int32_t u = ulo; u < uhi; ++u) {
        int32_t c = 0;
        for (int k = 0; k < T; ++k) c += j->all[k].cnt[u];
        j->counts[u] = c;
    }
    pthread_barrier_wait(j->bar);
    if (t == 0) { /* the prefix over U users is serial: 10^5 additions */
        j->offsets[0] = 0;
        for (int32_t u = 0; u < j->n_users; ++u) j->offsets[u + 1] = j->offsets[u] + j->counts[u];
        const size_t m_all = (size_t)j->offsets[j->n_users];
        *j->m_shared = m_all;
        if (m_all > 0 && m_all <= j->idx_cap) { /* bucket arrays sized by M, now that it is known */
            j->all[0].pairs = (pair_t *)malloc(m_all * sizeof(pair_t));
            j->all[0].tmp = (pair_t *)malloc(m_all * sizeof(pair_t));
            if (!j->all[0].pairs || !j->all[0].tmp) { free(j->all[0].pairs); free(j->all[0].tmp); j->all[0].pairs = j->all[0].tmp = NULL; *j->err = -3; }
        }
    }
    pthread_barrier_wait(j->bar);
    if (!j->all[0].pairs) return NULL; /* caller's idx too small, nothing selected, or out of memory: counts/offsets are done */
    /* phase 3: chunk k's first slot in bucket u = offsets[u] + rows of u in chunks before k (cnt becomes that slot,
     * relative to offsets[u]; it fits int32 because a bucket holds < 2^31 rows) */
    for (int32_t u = ulo; u < uhi; ++u) {
        int32_t run = 0;
        for (int k = 0; k < T; ++k) { const int32_t c = j->all[k].cnt[u]; j->all[k].cnt[u] = run; run += c; }
    }
    pthread_barrier_wait(j->bar);
    pair_t *pairs = j->all[0].pairs, *tmp = j->all[0].tmp;
    for (size_t k = 0; k < m; ++k) { /* fill buckets in row order */
        const int32_t i = sel[k];
        const int32_t u = j->user[i];
        pair_t *p = &pairs[j->offsets[u] + cnt[u]++];
        p->start = j->start[i];
        p->idx = i;
    }
    pthread_barrier_wait(j->bar);
    /* phase 4: per-bucket stable order, users split over the threads; scratch = the same slots of tmp */
    for (int32_t u = ulo; u < uhi; ++u)
        if (j->counts[u] > 1) stable_sort_pairs(pairs + j->offsets[u], tmp + j->offsets[u], (size_t)j->counts[u]);
    for (int64_t k = j->offsets[ulo]; k < j->offsets[uhi]; ++k) j->idx[k] = pairs[k].idx;
    return NULL;
}

int pie_oracle_scan_mt(const int64_t *start, const int64_t *end, const int32_t *user, const int32_t *disc, size_t n,
                       int32_t n_users, int64_t now, int64_t cutoff, uint64_t disc_mask, int32_t *counts,
                       int64_t *offsets, int32_t *idx, size_t idx_cap, size_t *m_out, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof(mt_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pthread_barrier_t bar;
    volatile int err = 0;
    size_t m = 0;
    if (!jobs || !th) { free(jobs); free(th); return -3; }
    pthread_barrier_init(&bar, NULL, (unsigned)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        mt_job *j = &jobs[t];
        j->start = start; j->end = end; j->user = user; j->disc = disc; j->n = n; j->n_users = n_users;
        j->now = now; j->cutoff = cutoff; j->mask = disc_mask; j->n_threads = n_threads; j->tid = t; j->all = jobs;
        j->counts = counts; j->offsets = offsets; j->idx = idx; j->bar = &bar;
        j->err = &err; j->m_shared = &m; j->idx_cap = idx_cap;
    }
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    for (int t = 0; t < n_threads; ++t) { free(jobs[t].sel); free(jobs[t].cnt); }
    free(jobs[0].pairs); free(jobs[0].tmp);
    free(jobs); free(th);
    if (m_out) *m_out = m;
    if (err) return err;
    return m > idx_cap ? -1 : 0;
}

typedef struct {
    uint64_t seed; int64_t n_total, row0, n; int32_t n_users, n_disc; uint32_t flags;
    int64_t *start, *end; int32_t *user, *disc;
} gen_job;

static void *gen_worker(void *arg)
{
    gen_job *g = (gen_job *)arg;
    pie_oracle_gen(g->seed, g->n_total, g->row0, g->n, g->n_users, g->n_disc, g->flags, g->start, g->end, g->user, g->disc);
    return NULL;
}

void pie_oracle_gen_mt(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                       uint32_t flags, int64_t *start, int64_t *end, int32_t *user, int32_t *disc, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    gen_job *jobs = (gen_job *)calloc((size_t)n_threads, sizeof(gen_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); pie_oracle_gen(seed, n_total, row0, n, n_users, n_disc, flags, start, end, user, disc); return; }
    for (int t = 0; t < n_threads; ++t) {
        const int64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        gen_job *g = &jobs[t];
        g->seed = seed; g->n_total = n_total; g->row0 = row0 + lo; g->n = hi - lo; g->n_users = n_users; g->n_disc = n_disc;
        g->flags = flags; g->start = start + lo; g->end = end + lo; g->user = user + lo; g->disc = disc + lo;
        pthread_create(&th[t], NULL, gen_worker, g);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(jobs); free(th);
}

/* ---------------------------------------------------------------- "next" row: dispatch-queue compaction */

int pie_oracle_expired_queue(const int64_t *end, size_t n, int64_t prev_now, int64_t now, int32_t *queue,
                             size_t cap, size_t *q_out)
{
    /* newly dead since the previous scan: dead at `now` (end <= now, server/sessionStore.js:69) and not yet dead
     * at `prev_now` (end > prev_now); queue keeps row order — the sequential-await order of
     * server/storage/sqlProvider.js:834-861 */
    size_t q = 0;
    for (size_t i = 0; i < n; ++i) {
        if (end[i] <= now && end[i] > prev_now) {
            if (q < cap) queue[q] = (int32_t)i;
            ++q;
        }
    }
    if (q_out) *q_out = q;
    return q > cap ? -1 : 0;
}

/* ---------------------------------------------------------------- "next" row: archive group-min chain */

int pie_oracle_archive_queue(const int64_t *start, const int64_t *end, const int32_t *user, size_t n, int32_t n_users,
                             int64_t now, int64_t window_ms, int32_t *queue, size_t cap, size_t *q_out)
{
    /* groups = new Map(); rows.forEach(...groups.get(key).push(...)) — server/storage/sqlProvider.js:763-782 */
    int64_t *earliest = (int64_t *)malloc((size_t)n_users * sizeof(int64_t));
    int32_t *order = (int32_t *)malloc((size_t)n_users * sizeof(int32_t)); /* keys in first-appearance order */
    unsigned char *seen = (unsigned char *)calloc((size_t)n_users, 1);
    if (!earliest || !order || !seen) { free(earliest); free(order); free(seen); return -3; }
    int32_t n_groups = 0;
    for (size_t i = 0; i < n; ++i) {
        if (end[i] == INT64_MIN) continue; /* deleted row: not in the table */
        const int32_t g = user[i];
        if (g < 0 || g >= n_users) { free(earliest); free(order); free(seen); return -2; }
        if (!seen[g]) { seen[g] = 1; order[n_groups++] = g; earliest[g] = start[i]; }
        else if (start[i] < earliest[g]) earliest[g] = start[i]; /* list.reduce(min), :785-794 */
    }
    /* now - earliest >= AUTO_ARCHIVE_WINDOW_MS (:798); JS numbers cannot overflow, int64 can: compare without
     * forming the difference when it would */
    unsigned char *qual = seen; /* reuse: 1 = seen, 2 = qualifies */
    for (int32_t k = 0; k < n_groups; ++k) {
        const int32_t g = order[k];
        const __int128 diff = (__int128)now - (__int128)earliest[g];
        if (diff >= (__int128)window_ms) qual[g] = 2;
    }
    /* for (const list of groups.values()) ... for (const item of list) archivedShows.push(...) — :784-811:
     * group order = first appearance, row order inside the group */
    size_t q = 0;
    for (int32_t k = 0; k < n_groups; ++k) {
        const int32_t g = order[k];
        if (qual[g] != 2) continue;
        for (size_t i = 0; i < n; ++i) { /* O(groups x n): this is an oracle for small cases */
            if (user[i] == g && end[i] != INT64_MIN) {
                if (q < cap) queue[q] = (int32_t)i;
                ++q;
            }
        }
    }
    free(earliest); free(order); free(seen);
    if (q_out) *q_out = q;
    return q > cap ? -1 : 0;
}

/* ---------------------------------------------------------------- "next" row: retention purge (calendar months) */

#define JS_DATE_MAX 8640000000000000LL

int64_t pie_oracle_add_months(int64_t ts, int32_t months, int64_t tz_offset_ms, int *is_nan)
{
    if (is_nan) *is_nan = 0;
    if (ts > JS_DATE_MAX || ts < -JS_DATE_MAX) return ts; /* invalid Date: returned as is, sqlProvider.js:1003-1005 */
    const __int128 local = (__int128)ts + tz_offset_ms;
    /* floor division: milliseconds of the (local) second, seconds since the epoch */
    int64_t secs = (int64_t)(local / 1000), ms = (int64_t)(local % 1000);
    if (ms < 0) { ms += 1000; secs -= 1; }
    time_t t = (time_t)secs;
    struct tm tm;
    if (!gmtime_r(&t, &tm)) { if (is_nan) *is_nan = 1; return 0; }
    tm.tm_mon += months; /* date.setMonth(date.getMonth() + months), :1007: timegm normalises month and day overflow */
    const time_t t2 = timegm(&tm);
    const __int128 out = (__int128)t2 * 1000 + ms - tz_offset_ms;
    if (out > JS_DATE_MAX || out < -JS_DATE_MAX) { if (is_nan) *is_nan = 1; return 0; } /* TimeClip -> NaN */
    return (int64_t)out;
}

/* The same under a REAL time zone, given as the transition table the host builds from the engine's zone rules
 * (sph-pie_amd/host/tzTable.js): off[0] applies before T[0], off[i + 1] from T[i] on.  ECMAScript semantics of
 * `d.setMonth(d.getMonth() + months)` (ECMA-262 Date.prototype.setMonth, LocalTime / UTC abstract operations):
 *   t      = LocalTime(ts)          = ts + offset in force at the UTC instant ts
 *   t2     = MakeDate(MakeDay(Year(t), Month(t) + months, Date(t)), TimeWithinDay(t))     (day overflow rolls over)
 *   result = TimeClip(UTC(t2)),  UTC(t2) = t2 - offset for the LOCAL time t2, where a local time that is skipped (clocks put
 *            forward) or repeated (clocks put back) "must be interpreted using the time zone offset before the transition".
 * Pinned by tests/golden/addmonths_zones.json (vectors from the JS engine's own Date under seven zones). */
static int64_t tz_offset_at_utc(const int64_t *T, const int64_t *off, int32_t n, int64_t t)
{
    int32_t k = 0;
    while (k < n && T[k] <= t) ++k; /* an oracle: linear */
    return off[k];
}

static int64_t tz_utc_from_local(const int64_t *T, const int64_t *off, int32_t n, int64_t tl)
{
    /* transition i happens, read on the OLD offset's clock, at local time T[i] + off[i] */
    int32_t k = 0;
    while (k < n && T[k] + off[k] <= tl) ++k;
    int64_t u = tl - off[k];
    if (k >= 1 && u < T[k - 1]) u = tl - off[k - 1]; /* skipped local time (clocks put forward): the offset before the transition */
    return u;
}

int64_t pie_oracle_add_months_tz(int64_t ts, int32_t months, const int64_t *T, const int64_t *off, int32_t n, int *is_nan)
{
    if (is_nan) *is_nan = 0;
    if (ts > JS_DATE_MAX || ts < -JS_DATE_MAX) return ts;
    const __int128 local = (__int128)ts + tz_offset_at_utc(T, off, n, ts);
    int64_t secs = (int64_t)(local / 1000), ms = (int64_t)(local % 1000);
    if (ms < 0) { ms += 1000; secs -= 1; }
    time_t t = (time_t)secs;
    struct tm tm;
    if (!gmtime_r(&t, &tm)) { if (is_nan) *is_nan = 1; return 0; }
    tm.tm_mon += months;
    const time_t t2 = timegm(&tm);
    const __int128 out_local = (__int128)t2 * 1000 + ms;
    if (out_local > 2 * (__int128)JS_DATE_MAX || out_local < -2 * (__int128)JS_DATE_MAX) { if (is_nan) *is_nan = 1; return 0; }
    const int64_t out = tz_utc_from_local(T, off, n, (int64_t)out_local);
    if (out > JS_DATE_MAX || out < -JS_DATE_MAX) { if (is_nan) *is_nan = 1; return 0; }
    return out;
}

int pie_oracle_retention_queue_tz(const int64_t *start, const int64_t *end, size_t n_rows, int64_t now, int32_t months,
                                  const int64_t *T, const int64_t *off, int32_t n, int32_t *queue, size_t cap, size_t *q_out)
{
    size_t q = 0;
    for (size_t i = 0; i < n_rows; ++i) {
        if (end[i] == INT64_MIN) continue;
        int nan = 0;
        const int64_t expiry = pie_oracle_add_months_tz(start[i], months, T, off, n, &nan);
        if (!nan && now >= expiry) {
            if (q < cap) queue[q] = (int32_t)i;
            ++q;
        }
    }
    if (q_out) *q_out = q;
    return q > cap ? -1 : 0;
}

int pie_oracle_retention_queue(const int64_t *start, const int64_t *end, size_t n, int64_t now, int32_t months,
                               int64_t tz_offset_ms, int32_t *queue, size_t cap, size_t *q_out)
{
    size_t q = 0;
    for (size_t i = 0; i < n; ++i) {
        if (end[i] == INT64_MIN) continue;
        int nan = 0;
        const int64_t expiry = pie_oracle_add_months(start[i], months, tz_offset_ms, &nan);
        if (!nan && now >= expiry) { /* _isArchiveExpired, :991-997 */
            if (q < cap) queue[q] = (int32_t)i;
            ++q;
        }
    }
    if (q_out) *q_out = q;
    return q > cap ? -1 : 0;
}
