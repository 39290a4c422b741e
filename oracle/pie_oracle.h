/*
 * TEST INFRASTRUCTURE — CPU oracle for the session-scan -> per-user feed path.
 *
 * This is a plain-C restatement of the predicates, ordering rule and bucket rule that the reference
 * (sphereisaiahmin-dev/sph-pie, mounted read-only at /root/reference) scatters over three files.  The
 * reference has NO aggregated scan function: the composition below is the [DERIVED]
 * contract of SURVEY.md §8(a-D).  Every predicate cites the reference line it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * path (libpie_hip.so) never links, loads or falls back to it.
 *
 * Parity pin: the liveness / purge / user-match predicates are pinned by tests/golden/sessionstore_g1_g4.json,
 * which holds outputs of the real server/sessionStore.js (see oracle/gen_golden.js).  The window, de-dup and
 * ordering rules have no executable reference here (calendarFeed.js / sqlProvider.js do not load on Node 12
 * and need absent packages): they are pinned by hand-derived vectors only — "parity unpinned" for those rows.
 */
#ifndef PIE_ORACLE_H
#define PIE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Synthetic corpus.  Row i consumes outputs 4i..4i+3 of the sequential splitmix64 stream
 * started at `seed`; rows [row0, row0+n) are written, so a table can be produced in slices. */
#define PIE_ORACLE_T0_MS        1700000000000LL
#define PIE_ORACLE_SPAN_MS      10368000000LL   /* 120 d */
#define PIE_ORACLE_TTL_MS       43200000LL      /* server/sessionStore.js:3 */
#define PIE_ORACLE_MIN_DUR_MS   900000LL        /* 15 min, "interval" variant */

#define PIE_GEN_INTERVAL   1u   /* end = start + uniform[15 min, 12 h] instead of start + TTL */
#define PIE_GEN_CLUSTERED  2u   /* user = floor(i*U/n_total): rows of one user are contiguous */
#define PIE_GEN_TIME_ORDERED 4u /* start = T0 - SPAN + 1 + floor(i*SPAN/n_total): rows in order of creation, as a session
                                   store appends them — the live rows sit together at the end of the table */

void pie_oracle_gen(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                    uint32_t flags, int64_t *start, int64_t *end, int32_t *user, int32_t *disc);

/* Skewed-user variant (SURVEY.md §8d "Zipf(1.1)"): like pie_oracle_gen, but user = the first k with r0 < cdf[k],
 * where cdf[0..n_users) are ascending 64-bit thresholds supplied by the caller (floor(CDF_k * 2^64), last = 2^64-1).
 * The table is computed once on the host in floating point and handed to both the oracle and the product, so the
 * corpus itself stays integer-exact on both sides. */
void pie_oracle_gen_cdf(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                        uint32_t flags, const uint64_t *cdf, int64_t *start, int64_t *end, int32_t *user, int32_t *disc);

/* Row-level predicate of the [DERIVED] contract. */
int pie_oracle_selected(int64_t start, int64_t end, int32_t disc, int64_t now, int64_t cutoff, uint64_t disc_mask);

/* Full scan: counts[U], offsets[U+1], idx[M].  Returns 0, or -1 if idx_cap < M (m_out still set), or -2 on a
 * user id outside [0, n_users). */
int pie_oracle_scan(const int64_t *start, const int64_t *end, const int32_t *user, const int32_t *disc, size_t n,
                    int32_t n_users, int64_t now, int64_t cutoff, uint64_t disc_mask, int32_t *counts,
                    int64_t *offsets, int32_t *idx, size_t idx_cap, size_t *m_out);

/* The same scan on `n_threads` host threads (baseline B2 "all host threads" of BASELINE.md section 3; bench.py's
 * cpu_baseline leg only).  Rows are split into contiguous chunks, one per thread: select + per-thread histogram, then a
 * per-user prefix over the threads (so every bucket is still filled in row order), then the buckets are ordered in
 * parallel over users.  Output identical to pie_oracle_scan.  Same return codes. */
int pie_oracle_scan_mt(const int64_t *start, const int64_t *end, const int32_t *user, const int32_t *disc, size_t n,
                       int32_t n_users, int64_t now, int64_t cutoff, uint64_t disc_mask, int32_t *counts,
                       int64_t *offsets, int32_t *idx, size_t idx_cap, size_t *m_out, int n_threads);
/* pie_oracle_gen on n_threads threads (row slices). */
void pie_oracle_gen_mt(uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users, int32_t n_disc,
                       uint32_t flags, int64_t *start, int64_t *end, int32_t *user, int32_t *disc, int n_threads);

/* "next" row: newly-expired change predicate  prev_now < end <= now  -> ordered queue. */
int pie_oracle_expired_queue(const int64_t *end, size_t n, int64_t prev_now, int64_t now, int32_t *queue,
                             size_t cap, size_t *q_out);

/* "next" row, the reference's own chain (server/storage/sqlProvider.js:758-816 _archiveDailyShows): rows are
 * grouped by a key (there: the show's date string; here [DERIVED]: the user column), a group's `earliest` is the
 * minimum createdAt (= start) of its rows, a group qualifies iff now - earliest >= window, and EVERY row of a
 * qualifying group is queued — groups in order of first appearance (Map insertion order, :769-789), rows in table
 * order inside a group — for the sequential dispatch of :834-861.  Tombstoned rows (end == INT64_MIN) are absent. */
int pie_oracle_archive_queue(const int64_t *start, const int64_t *end, const int32_t *user, size_t n, int32_t n_users,
                             int64_t now, int64_t window_ms, int32_t *queue, size_t cap, size_t *q_out);

/* "next" row: retention purge with calendar-month arithmetic, server/storage/sqlProvider.js:991-1009.
 * expiry = `date.setMonth(date.getMonth() + months)` on a local-time Date; here local = UTC + tz_offset_ms (a fixed
 * offset; DST is not modelled).  JS Date range rules apply: |ts| > 8.64e15 is an invalid Date and is returned unchanged
 * (:1003-1005); a result outside the range is NaN (*is_nan = 1) and `now >= NaN` is false (:996).
 * This restatement goes through libc (gmtime_r / timegm) — a different route from the product's integer civil-date
 * arithmetic — and is pinned by tests/golden/addmonths_utc.json (vectors from the JS engine's own Date). */
int64_t pie_oracle_add_months(int64_t ts, int32_t months, int64_t tz_offset_ms, int *is_nan);
/* the same under a real time zone given as a transition table (off[0] before T[0], off[i + 1] from T[i] on; ms): ECMA-262
 * setMonth on a local Date, skipped / repeated local times read with the offset before the transition; pinned by
 * tests/golden/addmonths_zones.json */
int64_t pie_oracle_add_months_tz(int64_t ts, int32_t months, const int64_t *T, const int64_t *off, int32_t n, int *is_nan);
int pie_oracle_retention_queue_tz(const int64_t *start, const int64_t *end, size_t n_rows, int64_t now, int32_t months,
                                  const int64_t *T, const int64_t *off, int32_t n, int32_t *queue, size_t cap, size_t *q_out);
/* rows (not tombstoned) with now >= addMonths(start, months), ascending row order (:863-890 _purgeExpiredArchives) */
int pie_oracle_retention_queue(const int64_t *start, const int64_t *end, size_t n, int64_t now, int32_t months,
                               int64_t tz_offset_ms, int32_t *queue, size_t cap, size_t *q_out);

/* user-hash sharding rule shared with the product: rank = splitmix64(user) mod G. */
uint64_t pie_oracle_splitmix64(uint64_t x);
int32_t pie_oracle_shard_of(int32_t user, int32_t n_shards);

#ifdef __cplusplus
}
#endif
#endif
