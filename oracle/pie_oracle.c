/*
 * TEST INFRASTRUCTURE — CPU oracle (see pie_oracle.h for scope and the parity-pin statement).
 * Citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "pie_oracle.h"

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
        const int64_t s = PIE_ORACLE_T0_MS - (int64_t)mulhi64(r2, (uint64_t)PIE_ORACLE_SPAN_MS);
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
