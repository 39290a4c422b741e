/*
 * pie_scan.h — C ABI of the MI355X session-scan -> per-user feed path (libpie_hip.so).
 *
 * The reference (sphereisaiahmin-dev/sph-pie) is a pure Node.js app with no FFI seam; the seams this ABI
 * sits behind are the CommonJS exports of three modules and one HTTP route.  Each entry
 * point below names the reference interface it replaces (paths relative to /root/reference).  The Node
 * binding (raw N-API, sph-pie_amd/csrc/pie_napi.c) and the ctypes binding (sph-pie_amd/binding.py) wrap
 * exactly these symbols; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain pointers and sizes, caller-owned host buffers, no exceptions across the boundary,
 * int status return (0 = ok, negative = PIE_E_*), text of the last failure via pie_last_error().
 * A context is bound to one GPU and is used from one host thread at a time (the reference is a
 * single-threaded event loop: server/index.js, no worker threads).  There is NO CPU fallback: without a
 * HIP device pie_ctx_create() fails with PIE_E_NODEVICE.
 */
#ifndef PIE_SCAN_H
#define PIE_SCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIE_ABI_VERSION 1

enum {
    PIE_OK = 0,
    PIE_E_INVAL = -1,    /* bad argument (null pointer, negative size, n >= 2^31, user id out of range) */
    PIE_E_NODEVICE = -2, /* no HIP device / HIP runtime error at init */
    PIE_E_HIP = -3,      /* HIP runtime error during the call; see pie_last_error() */
    PIE_E_NOMEM = -4,    /* device or host allocation failed */
    PIE_E_CAPACITY = -5, /* caller buffer too small (idx_cap < M): *m_out holds the needed size */
    PIE_E_STATE = -6     /* call out of order (scan before load, fetch before scan, ...) */
};

/* synthetic-corpus flags; identical meaning in oracle/pie_oracle.h */
#define PIE_GEN_INTERVAL  1u /* end = start + uniform[15 min, 12 h]; default end = start + SESSION_TTL_MS */
#define PIE_GEN_CLUSTERED 2u /* rows of one user contiguous; default uniform random users */
#define PIE_GEN_TIME_ORDERED 4u /* rows in order of creation (start ascending with the row index), as a session store appends
                                   them: the live rows sit together at the end of the table; default random order */

/* sentinel for "no end" (calendarFeed.js:74 endTs === null) and for tombstoned rows: never live */
#define PIE_END_NONE INT64_MIN

typedef struct pie_ctx pie_ctx;

typedef struct pie_stats {
    uint32_t struct_size;  /* set by caller to sizeof(pie_stats) */
    uint32_t n_profiled;   /* scans whose events were resolved into the sums below */
    uint64_t rows;         /* N of the resident table */
    uint64_t users;        /* U */
    uint64_t selected;     /* M of the last scan */
    uint64_t alg_bytes;    /* 24 * N: algorithmic bytes of one scan */
    double k1_ms_sum;      /* sum of predicate+compaction kernel durations (HIP events, scan stream) */
    double scan_ms_sum;    /* sum of first-kernel-start -> last-kernel-end durations */
    uint32_t max_bucket;   /* largest per-user bucket of the last scan */
    uint32_t n_segments;   /* block-sorted segments of the last scan */
    uint32_t n_big;        /* buckets that needed the multi-pass merge in the last scan */
    uint32_t k1_blocks;    /* grid of the scan kernel */
    uint32_t k1_variant;   /* form of the scan kernel used by the last scan (bit0 nt loads, bit1 late user, bit2 liveness-first,
                              0x400 keyed: streams the 2-byte liveness key instead of the `end` column, 0x800 the 1-byte key,
                              0x1000 batched: the last finished call was a batch, `selected` sums its queries, `max_bucket` is the
                              largest UNION bucket,
                              0x2000 ordered run: 0x2003 dense form, 0x2400 / 0x2C00 keyed form on the 2- / 1-byte key,
                              0x3400 / 0x3C00 a batch on the ordered run) */
    uint32_t key_ambiguous; /* keyed form: rows of the last scan whose key equalled the query's and needed the full compare (saturating) */
    uint64_t live;         /* rows with end > now seen by the last scan */
    uint64_t candidates;   /* keyed form: rows whose key was >= the query's, i.e. payload records the table pass gathered */
} pie_stats;

/* ---- lifecycle ------------------------------------------------------------------------------------- */
int pie_abi_version(void);
int pie_device_count(void);
/* One context per GPU, one process per GPU.  device_id is the HIP ordinal. */
int pie_ctx_create(int device_id, pie_ctx **ctx_out);
int pie_ctx_destroy(pie_ctx *ctx);
/* ctx may be NULL: returns the last error of a failed pie_ctx_create on this thread. */
const char *pie_last_error(const pie_ctx *ctx);
/* Run the scan's main-stream kernels (K1, K2) on a caller stream (a hipStream_t, e.g. torch's current stream).
 * NULL = ctx-owned stream. */
int pie_ctx_set_stream(pie_ctx *ctx, void *hip_stream);
/* The stream on which scan results are produced and on which every result copy / pack is enqueued (today the
 * context's one stream).  A caller that consumes results on another stream records an event here
 * (e.g. torch.cuda.ExternalStream(ptr)). */
int pie_ctx_aux_stream(pie_ctx *ctx, void **hip_stream_out);

/* ---- session table: replaces the in-process `sessions` Map (server/sessionStore.js:6,17) -------------
 * Columns are SoA: start = createdAt, end = expiresAt (int64 ms), user = dense index of the userId string,
 * disc = index into DISCIPLINES (server/disciplineConfig.js:35).  Host arrays stay caller-owned. */
int pie_load_columns(pie_ctx *ctx, const int64_t *start, const int64_t *end, const int32_t *user,
                     const int32_t *disc, size_t n, int32_t n_users);
/* createSession (server/sessionStore.js:12-19): append k rows behind the resident ones (device capacity grows
 * geometrically).  n_users may grow, never shrink. */
int pie_append_rows(pie_ctx *ctx, const int64_t *start, const int64_t *end, const int32_t *user, const int32_t *disc,
                    size_t k, int32_t n_users);
/* Fill the table on the device with rows [row0, row0+n) of the deterministic synthetic corpus. */
int pie_gen_synthetic(pie_ctx *ctx, uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users,
                      int32_t n_disc, uint32_t flags);
/* Same corpus with skewed users (the "Zipf(1.1)" variant of SURVEY.md §8d): user = first k with r0 < cdf[k], cdf =
 * n_users ascending 64-bit thresholds floor(CDF_k * 2^64) computed by the caller. */
int pie_gen_synthetic_cdf(pie_ctx *ctx, uint64_t seed, int64_t n_total, int64_t row0, int64_t n, int32_t n_users,
                          int32_t n_disc, uint32_t flags, const uint64_t *cdf);
/* Flat column files: <dir>/{start.i64,end.i64,user.i32,disc.i32} + header.json; the table survives a
 * restart (the reference keeps sessions in memory only, server/sessionStore.js:6).  Load mmaps the files and uploads. */
int pie_save_columns(pie_ctx *ctx, const char *dir);
int pie_load_columns_dir(pie_ctx *ctx, const char *dir);
/* Copy the resident columns back (any pointer may be NULL). */
int pie_read_columns(pie_ctx *ctx, int64_t *start, int64_t *end, int32_t *user, int32_t *disc, size_t n);
/* touchSession (server/sessionStore.js:37-45): end[row] = new_end.  deleteSession (:47-53): new_end = PIE_END_NONE. */
int pie_set_end(pie_ctx *ctx, const int32_t *rows, const int64_t *new_end, size_t k);
/* deleteSessionsForUser (server/sessionStore.js:55-64): tombstone every live row with user == u (strict match;
 * unknown ids are a no-op like the falsy-id guard :56-58).  rows_out (may be NULL) receives the tombstoned row
 * indices in ascending order so the host can drop their token-map entries; *n_deleted their number. */
int pie_delete_user(pie_ctx *ctx, int32_t user, int32_t *rows_out, size_t cap, size_t *n_deleted);

/* _pruneCalendarEvents (server/storage/sqlProvider.js:956-968): tombstone every row with start < cutoff (the
 * complement of the scan's window predicate); rows_out / n_pruned as in pie_delete_user. */
int pie_prune_before(pie_ctx *ctx, int64_t cutoff, int32_t *rows_out, size_t cap, size_t *n_pruned);

/* Retention purge with calendar-month arithmetic (server/storage/sqlProvider.js:863-890 _purgeExpiredArchives, :991-1009
 * _isArchiveExpired / _addMonths; ARCHIVE_RETENTION_MONTHS = 2, :10): tombstone every row with
 * now >= addMonths(start, months), where addMonths is JS `setMonth(getMonth() + months)` on a local-time Date (day
 * overflow rolls into the next month) and local time = UTC + tz_offset_ms (a fixed offset of whole minutes; zones with
 * daylight saving: pie_retention_purge_tz).  rows_out / n_purged as in pie_delete_user. */
int pie_retention_purge(pie_ctx *ctx, int64_t now, int32_t months, int64_t tz_offset_ms, int32_t *rows_out, size_t cap,
                        size_t *n_purged);
/* The same under a REAL time zone (the reference's Date is local: daylight saving moves the result by the hour the clocks
 * move).  The zone travels as the transition table the host builds from its own zone rules (sph-pie_amd/host/tzTable.js,
 * binding.tz_table): offsets_ms[0] applies before transitions_utc_ms[0], offsets_ms[i + 1] from transitions_utc_ms[i] on
 * (n_transitions + 1 offsets; local = UTC + offset).  The device follows ECMA-262 exactly: offset in force at the UTC instant,
 * month shift on the local fields, and a resulting local time that is skipped or repeated at a transition is read with the
 * offset before the transition.  Instants outside the table's span use its first / last offset.  n_transitions = 0 is the
 * fixed-offset form.  Pinned by tests/golden/addmonths_zones.json (JS engine vectors under seven zones). */
int pie_retention_purge_tz(pie_ctx *ctx, int64_t now, int32_t months, const int64_t *transitions_utc_ms, const int64_t *offsets_ms,
                           int32_t n_transitions, int32_t *rows_out, size_t cap, size_t *n_purged);

/* ---- discipline predicate table: replaces findDiscipline() lookups (server/disciplineConfig.js:88-97) -
 * bit d of mask = rows of discipline d are wanted; bits >= n_disc are ignored. n_disc <= 64. */
int pie_set_disciplines(pie_ctx *ctx, uint64_t mask, int32_t n_disc);

/* ---- the scan: replaces the per-request loops (server/sessionStore.js:59-63,68-72;
 * server/storage/sqlProvider.js:284 window filter, :276 ORDER BY start_ts ASC) -------------------------
 * Row i is selected iff end[i] > now && start[i] >= cutoff && bit(mask, disc[i]).  Feed(u) = selected rows
 * of user u ordered by (start asc, row index asc).  Outputs: counts[U], offsets[U+1], idx[M]. */
int pie_scan(pie_ctx *ctx, int64_t now, int64_t cutoff, int32_t *counts_out, int64_t *offsets_out,
             int32_t *idx_out, size_t idx_cap, size_t *m_out);
/* Same scan, results left in device memory (for the multi-GPU gather and for benchmarking). */
int pie_scan_device(pie_ctx *ctx, int64_t now, int64_t cutoff, size_t *m_out);
/* The same scan in two halves for callers that overlap host work with it: begin enqueues the table pass and returns at
 * once; finish waits for the scan's summary (M), enqueues what is left (scatter + order of buckets that outgrew their
 * direct slots, the rare big-bucket merge passes) and returns M.  Up to two scans may be in flight: with begin(i+1)
 * called before finish(i), the offsets + order kernel of scan i runs inside the launch of scan i+1's table pass, so
 * a steady stream of scans costs one kernel launch each — and the summary of scan i arrives while scan i+1 runs (a
 * caller that wants it sooner calls finish(i) first).  Results of a finished scan end at the next begin. */
int pie_scan_begin(pie_ctx *ctx, int64_t now, int64_t cutoff);
int pie_scan_finish(pie_ctx *ctx, size_t *m_out);
/* The same pair for the exchange step of a sharded table: the scan also produces its result message
 * [ off[0..u_pad] | M | rows[0..min(M, idx_cap)) ] (int32 words, the layout of pie_pack_results_device) in caller-owned
 * device memory dst_i32 (u_pad + 2 + idx_cap words, which must stay valid until the matching finish).
 * pie_scan_finish_packed: *ready_out = 1 when the message was complete in device memory before the call returned (the
 * scan's own kernels wrote it: a consumer on any stream, or a peer GPU, may read it at once); 0 when a pack kernel was
 * enqueued on the context's stream to write it (order the consumer after that stream, e.g. with an event). */
int pie_scan_begin_packed(pie_ctx *ctx, int64_t now, int64_t cutoff, void *dst_i32, size_t u_pad, size_t idx_cap);
int pie_scan_finish_packed(pie_ctx *ctx, size_t *m_out, int *ready_out);
/* pie_scan_begin_packed with a second destination: counts_dst_i32 (n_users words, may be NULL) receives counts[U].  Both
 * destinations are device-visible memory; they may be MAPPED PINNED HOST memory (pie_host_alloc): the scan's own kernels
 * then deliver offsets / counts / rows to the host with no copy node behind the scan ("D2H of
 * counts/offsets included"), complete when pie_scan_finish_packed returns with *ready_out = 1 or, with 0, after
 * pie_synchronize. */
int pie_scan_begin_packed2(pie_ctx *ctx, int64_t now, int64_t cutoff, void *dst_i32, size_t u_pad, size_t idx_cap,
                           void *counts_dst_i32);
/* Pinned host memory mapped into the device's address space: *host_out is the CPU address, *dev_out the address a
 * kernel (or pie_scan_begin_packed*) uses for the same bytes. */
int pie_host_alloc(pie_ctx *ctx, size_t bytes, void **host_out, void **dev_out);
int pie_host_free(pie_ctx *ctx, void *host_ptr);
/* ---- batched scan: many feed requests, one table pass (SURVEY.md section 7 "batch many queries per launch"; the
 * north_star's "calendarFeed per-request loop -> batched GPU scan").  Every query has its own `now` (the request's clock,
 * server/sessionStore.js:67 samples one per scan), `cutoff` (server/calendarFeed.js:33-38) and discipline mask
 * (server/disciplineConfig.js:88-97; bits >= n_disc of pie_set_disciplines are ignored).  Up to three batches per lane (see
 * pie_set_batch_lanes) may be in flight (begin(i+1), begin(i+2) before finish(i): the tail of batch i rides in the launch of the
 * next batch of its lane, and with a third batch queued the GPU never waits for the host to react to a summary); single scans
 * and batches do not mix in flight.
 *
 * The PRIMARY result of a batch is the UNION of its queries' selections: per user the rows that ANY query selected, in
 * (start, row) order, with a query mask per row —
 *     uoff[U+1] (int64) | rows[Mu] (int32) | mask[Mu] (bit q = query q selected the row)
 *     Feed(q, u) = the rows of rows[uoff[u] : uoff[u+1]] whose mask has bit q, in that order
 * (requests that arrive together select almost the same rows: the union is little longer than one query's list, whatever Q).
 * pie_scan_batch_finish reports every query's M; pie_batch_read_user_feed answers a request straight from the union.  The
 * per-query form of a result — counts[U], offsets[U+1], idx[M], bit for bit those of n_q separate pie_scan calls — is
 * materialised from the union only when asked for (pie_batch_read_results, pie_batch_result_device_ptrs, the per-query
 * messages).  A query the batched pass cannot hold (a dense query, a user with more than 64 union rows) is rerun inside
 * pie_scan_batch_finish on the general path; such a batch has per-query results only (no union: pie_batch_union_device_ptrs
 * returns NULL pointers). */
#define PIE_BATCH_MAX 64
typedef struct pie_query {
    int64_t now, cutoff;
    uint64_t mask;
} pie_query;
int pie_scan_batch_begin(pie_ctx *ctx, const pie_query *queries, int n_q);
/* m_out: n_q selected-row counts (may be NULL) */
int pie_scan_batch_finish(pie_ctx *ctx, size_t *m_out);
/* begin + finish */
int pie_scan_batch(pie_ctx *ctx, const pie_query *queries, int n_q, size_t *m_out);
/* The union of the last finished batch: device pointers (valid until three more batches have begun; NULL when the batch has
 * no union, see above), or host copies (masks_out: one 64-bit mask per union row; PIE_E_CAPACITY if cap < Mu, PIE_E_STATE if
 * the batch has no union). */
int pie_batch_union_device_ptrs(pie_ctx *ctx, void **uoff_dev /* int64[U+1] */, void **rows_dev /* int32[Mu] */,
                                void **mask_lo_dev /* uint32[Mu]: queries 0..31 */, void **mask_hi_dev /* uint32[Mu]: 32..63, NULL for n_q <= 32 */,
                                size_t *mu_out);
int pie_batch_read_union(pie_ctx *ctx, int64_t *uoff_out, int32_t *rows_out, uint64_t *masks_out, size_t cap, size_t *mu_out);
/* Batch LANES.  A batch over a shard-sized table (a tenth of 10^8 rows) is one launch of ~20 us that occupies a fraction of
 * the chip: its time is latency, not bytes, and that floor is what would cap an 8-GPU split of the table at 2.5x.  A context
 * therefore deals its batches to up to four lanes — independent pipelines, each with its own HIP stream, three batch slots and
 * spans — whose launches run SIDE BY SIDE on the chip; pie_scan_batch_begin picks the lane (round robin),
 * pie_scan_batch_finish returns batches in the order they were begun, so callers do not change: with n lanes up to 3 n batches
 * may be in flight.  n_lanes 1..4 pins the number, 0 (default; PIE_BATCH_LANES) chooses by table size: 4 up to 2^25 rows, 3
 * above.  Batches on the ordered run and batches that only fall back stay on lane 0.  Reading a finished batch's arrays
 * on the context's stream is ordered behind the lane by the library; the messages of pie_scan_batch_begin_union are complete
 * when finish says ready, as before.  There is no counterpart in the reference (one request at a time,
 * server/index.js:293-302). */
int pie_set_batch_lanes(pie_ctx *ctx, int n_lanes);
int pie_batch_lanes(pie_ctx *ctx); /* lanes in use now */
/* Batches pie_scan_batch_begin would take right now (0: finish one first): three per lane less those in flight; a table whose
 * batches run on the ordered run uses lane 0 only, whatever the lane count.  A pipelined caller asks this instead of counting. */
int pie_batch_room(pie_ctx *ctx);
/* The end of a burst: no further begin is coming for now.  The tail of a batch normally rides in its lane's next launch and,
 * for a lane's last batch, is queued when that batch is finished — one after the other as the caller works through them.
 * pie_scan_batch_flush queues the waiting tails of all lanes at once (they run side by side); optional, results unchanged; a
 * begin after it simply carries no tail. */
int pie_scan_batch_flush(pie_ctx *ctx);
/* A batch that also writes the multi-GPU exchange message as it goes — ONE union message for the whole batch:
 *   msg (int32 words) = [ uoff[0..u_pad] | Mu | rows[0..cap) | mask_lo[0..cap) | mask_hi[0..cap) (only when n_q > 32) ]
 * u_pad + 2 + 2 * cap words (3 * cap for n_q > 32); uoff[u] = Mu for u >= users; rows beyond cap are dropped (Mu says how many
 * there are).  msg is device-visible memory (device or mapped host) that stays valid until the matching finish.
 * pie_scan_batch_finish_packed: *ready_out = 1 when the message was complete when the call returned (the batch's own kernels
 * wrote it); 0 when it was packed afterwards on the context's stream (order the consumer behind pie_ctx_aux_stream) — then
 * Mu = -1 means a user's merged union exceeds 32 rows or the batch holds more than 32 queries: use the per-query messages. */
int pie_scan_batch_begin_union(pie_ctx *ctx, const pie_query *queries, int n_q, void *msg_i32, size_t u_pad, size_t cap);
/* The per-query form of the exchange: one result message per query (layout of pie_pack_results_device) written to
 * msg_i32 + q * msg_stride_words and, optionally, counts[U] to counts_i32 + q * counts_stride_words; both device-visible.
 * The lists are materialised from the union and packed at finish (*ready_out = 0: order the consumer behind the context's
 * stream). */
int pie_scan_batch_begin_packed(pie_ctx *ctx, const pie_query *queries, int n_q, void *msg_i32, size_t msg_stride_words,
                                size_t u_pad, size_t idx_cap, void *counts_i32, size_t counts_stride_words);
int pie_scan_batch_finish_packed(pie_ctx *ctx, size_t *m_out, int *ready_out);
/* Results of query `qi` of the last finished batch (as pie_read_results / pie_result_device_ptrs): materialised on first use. */
int pie_batch_read_results(pie_ctx *ctx, int qi, int32_t *counts_out, int64_t *offsets_out, int32_t *idx_out, size_t idx_cap,
                           size_t *m_out);
int pie_batch_result_device_ptrs(pie_ctx *ctx, int qi, void **counts_dev, void **offsets_dev, void **idx_dev);
/* The union message (layout above) of the last finished batch into caller memory, whatever path the batch took; enqueued on
 * the context's stream (pie_ctx_aux_stream). */
int pie_batch_pack_union_device(pie_ctx *ctx, void *dst_i32, size_t u_pad, size_t cap);
/* One user's feed of query `qi` of the last finished batch (as pie_read_user_feed): the per-request read of a server that
 * answers the requests of one event-loop turn with one batch (/root/reference/server/index.js:293-302).  Read from the union
 * (two small copies + a filter on the host); no per-query list is built for it. */
int pie_batch_read_user_feed(pie_ctx *ctx, int qi, int32_t user, int32_t *idx_out, size_t idx_cap, size_t *k_out);

/* The requests of one event-loop turn, fetched together: request i asks for Feed(qi[i], user[i]) of the last finished batch.
 * req_off_out[n_req + 1] (exclusive offsets into the arrays below), then for every request its rows in feed order and the
 * columns the host serialises (start, end, disc: the event object of server/calendarFeed.js:66-79) — two device round trips
 * for the whole batch instead of three small copies per request.  A user outside [0, U) has an empty feed.  PIE_E_CAPACITY if
 * the feeds hold more than cap_rows rows (*total_out says how many). */
int pie_batch_fetch_requests(pie_ctx *ctx, const int32_t *qi, const int32_t *user, size_t n_req, size_t cap_rows, int64_t *req_off_out,
                             int32_t *idx_out, int64_t *start_out, int64_t *end_out, int32_t *disc_out, size_t *total_out);

/* Copy the last finished scan's results to host arrays (what pie_scan does after scanning); any pointer may be NULL. */
int pie_read_results(pie_ctx *ctx, int32_t *counts_out, int64_t *offsets_out, int32_t *idx_out, size_t idx_cap,
                     size_t *m_out);
/* One user's feed of the last finished scan: rows idx[offsets[user] .. offsets[user+1]) into idx_out (two small copies
 * instead of the whole result: the per-request read behind GET /api/calendar, /root/reference/server/index.js:293-302).
 * *k_out = the feed's length; PIE_E_CAPACITY if it exceeds idx_cap; a user outside [0, U) has an empty feed. */
int pie_read_user_feed(pie_ctx *ctx, int32_t user, int32_t *idx_out, size_t idx_cap, size_t *k_out);
/* Device pointers of the last finished scan's results: complete in stream order (pie_ctx_aux_stream) or after
 * pie_synchronize; valid until the next pie_scan_begin. */
int pie_result_device_ptrs(pie_ctx *ctx, void **counts_dev, void **offsets_dev, void **idx_dev);
/* Copy the last scan's results into caller-owned DEVICE buffers (e.g. torch tensors that feed an RCCL
 * all-gather), asynchronously on the context's stream.  Any pointer may be NULL; idx copies min(M, idx_cap). */
int pie_copy_results_device(pie_ctx *ctx, void *counts_dst, void *offsets_dst, void *idx_dst, size_t idx_cap);
/* Pack the last scan's results into ONE int32 message in caller-owned device memory, one launch on the context's
 * stream: [off[0..u_pad] (exclusive offsets, = M past the last user) | M | idx[0 .. min(M, idx_cap))], u_pad+2+cap
 * words — the unit of the multi-GPU all-gather; Feed(u) = idx[off[u] : off[u+1]]. */
int pie_pack_results_device(pie_ctx *ctx, void *dst_i32, size_t u_pad, size_t idx_cap);
/* Gather the rows named by idx (host array of m row indices) for host-side serialisation
 * (the event object of server/calendarFeed.js:66-79).  Output pointers may be NULL. */
int pie_fetch_rows(pie_ctx *ctx, const int32_t *idx, size_t m, int64_t *start, int64_t *end, int32_t *user,
                   int32_t *disc);

/* ---- "next" row: newly-expired change predicate -> ordered dispatch queue -----------
 * queue = ascending row indices with prev_now < end <= now  (dead per server/sessionStore.js:69 at `now`,
 * not yet dead at `prev_now`); order = the sequential-await order of server/storage/sqlProvider.js:834-861. */
int pie_expired_queue(pie_ctx *ctx, int64_t prev_now, int64_t now, int32_t *queue_out, size_t cap, size_t *q_out);

/* The reference's own archive chain (server/storage/sqlProvider.js:758-816 _archiveDailyShows), [DERIVED] onto the
 * session table with the user column as the group key: a group's earliest = min(start) over its (non-tombstoned)
 * rows; it qualifies iff now - earliest >= window_ms (:798, AUTO_ARCHIVE_WINDOW_MS :9); every row of a qualifying
 * group is queued, groups in order of first appearance (Map insertion order), rows in table order inside a group —
 * the order in which :834-861 dispatches them.  Entirely on the device: one pass for the group statistics, the threshold and
 * the first-appearance ranks per group, an order-preserving selection pass, a stable sort of the queued rows by group rank. */
int pie_archive_queue(pie_ctx *ctx, int64_t now, int64_t window_ms, int32_t *queue_out, size_t cap, size_t *q_out);
/* Measurement of the archive chain: with pie_set_profiling on, the device time (first kernel start -> end of the last sort) of
 * the chains since pie_stats_reset and their number; the algorithmic bytes of the last one: 20 B/row for the group statistics
 * (start, end, user) + 12 B/row for the selection (end, user) + 4 B per queued row. */
int pie_archive_stats(pie_ctx *ctx, double *ms_sum_out, uint32_t *calls_out, uint64_t *alg_bytes_out);

/* ---- measurement ------------------------------------------------------------------------------------- */
/* Pin the form of the table pass (the codes of pie_stats.k1_variant, DESIGN.md section 8: 0x01 reads every byte of the
 * four columns, 0x03 the streaming form, 0xC85 the keyed form ...); form < 0 returns to the adaptive choice.  A tuning /
 * measurement switch: every form produces identical results.  Not while a scan is in flight. */
int pie_set_scan_form(pie_ctx *ctx, int form);
typedef struct pie_table_info {
    uint32_t struct_size;     /* set by caller to sizeof(pie_table_info) */
    uint32_t has_keys;        /* 1: the derived liveness-key / payload columns exist and are in step */
    uint64_t rows, users;
    uint64_t table_bytes;     /* 24 * capacity rows: the four caller-visible columns */
    uint64_t derived_bytes;   /* derived columns (2-byte key, 1-byte key, 16-byte payload record) */
    uint64_t workspace_bytes; /* per-scan workspace of the two slots + histogram spans */
    double index_build_ms;    /* host wall time of the last full build of the derived columns (kernels + syncs) */
    uint64_t ordered_rows;    /* rows the ordered run holds (0: there is none, or it was invalidated) */
    uint64_t ordered_bytes;   /* device memory of the ordered run (27 B per position + 4 B per row of capacity + small arrays) */
    double ordered_build_ms;  /* host wall time of its last build (one all-selecting scan + a gather) */
    uint64_t ordered_builds;  /* times it was built for this context */
    uint64_t ordered_positions; /* positions a scan of the run visits: its rows + the spare slots of every user's segment */
    uint64_t ordered_respreads; /* times appends filled a segment and the run was moved into fresh segments (a linear pass) */
} pie_table_info;
int pie_table_info_get(pie_ctx *ctx, pie_table_info *out);
/* The ordered run (sph-pie_amd/csrc/pie_ordered.h): the table's rows a second time, in (user, start, row) order — the order
 * of every answer — so that a query is a filter over positions: no histogram atomics, no per-bucket sort, no dependence on
 * how rows are spread over users.  There is no counterpart in the reference (its Map is scanned per request,
 * server/sessionStore.js:55-73); results are identical to the general path's.  mode 0: never (frees it); 1 (default):
 * built and used when the general path is weak — a query selecting more than 1/24 of the rows, or skewed users — the second
 * time in a row such a query arrives; 2: always (built at the next scan).  Touches and deletes keep it in step, and so do
 * appends: a new row is inserted at its place in its user's segment, which ends in spare slots — for a session store's
 * createSession that place is the end; a row a little late shifts the few behind it; a full segment moves the run into
 * fresh segments (a linear pass).  Loads, sharding and back-fills (a row more than 256 rows back in its segment) invalidate
 * it (queries run on the general path until it is rebuilt).  PIE_ORDERED=0|1|2 sets the mode a context starts with. */
int pie_set_ordered_run(pie_ctx *ctx, int mode);
/* 0: off.  n >= 1: every n-th scan carries HIP events around K1 and around the whole scan (an event between two
 * kernels costs a few microseconds of pipeline drain, so a benchmark samples). */
int pie_set_profiling(pie_ctx *ctx, int enabled);
int pie_stats_get(pie_ctx *ctx, pie_stats *out);  /* resolves pending events (synchronises the stream) */
int pie_stats_reset(pie_ctx *ctx);
int pie_synchronize(pie_ctx *ctx);
/* user-hash sharding rule: shard = splitmix64(user) mod n_shards.  Pure host function. */
int32_t pie_shard_of(int32_t user, int32_t n_shards);
/* Shard the RESIDENT table on the device: keep only the rows whose user hashes to `rank` of `world` (pie_shard_of), in table
 * order, users re-numbered densely in ascending global id (a shard with no user keeps n_users = 1).  Every rank loads or
 * generates the same whole table and calls this with its own rank; nothing crosses PCIe.  *n_rows_out / *n_users_out = the
 * shard's size.  pie_shard_maps copies the maps back: rows_global_out[i] = the global row of local row i (n_rows),
 * users_global_out[k] = the global id of local user k (n_users); either may be NULL. */
int pie_shard_table(pie_ctx *ctx, int32_t rank, int32_t world, size_t *n_rows_out, int32_t *n_users_out);
int pie_shard_maps(pie_ctx *ctx, int32_t *rows_global_out, int32_t *users_global_out);

/* ---- communicator: the sharded table behind the C ABI (SURVEY.md 8b "pie_ctx_create(device_ids[], n, ...)", 8e) --------
 * A pie_comm owns one scan context per GPU and one RCCL communicator; the session table is sharded by user hash
 * (pie_shard_of) and the cross-user reassembly — every rank receives every rank's per-user offsets and row lists — is an
 * all-gather done as grouped ncclSend / ncclRecv over the point-to-point xGMI links.  This is what lets a host that is not
 * Python (the Node addon) use more than one GPU.  RCCL is opened at run time; without it pie_comm_create fails with
 * PIE_E_NODEVICE.  Two ways to build one:
 *   pie_comm_create        one process drives all n GPUs of the node (ncclCommInitAll): the Node host;
 *   pie_comm_create_rank   one process per GPU (ncclCommInitRank): rank 0 makes the 128-byte id with pie_comm_unique_id
 *                          and hands it to the other ranks by whatever channel the host has. */
typedef struct pie_comm pie_comm;
int pie_comm_create(const int32_t *device_ids, int32_t n, pie_comm **comm_out);
int pie_comm_unique_id(void *id_out_128);
int pie_comm_create_rank(const void *id_128, int32_t rank, int32_t world, int32_t device_id, pie_comm **comm_out);
int pie_comm_destroy(pie_comm *comm);
/* comm may be NULL: the last error of a failed pie_comm_create* on this thread */
const char *pie_comm_last_error(const pie_comm *comm);
int32_t pie_comm_world(const pie_comm *comm);
int32_t pie_comm_local_ranks(const pie_comm *comm);
/* The scan context of shard `rank` (NULL when that rank lives in another process): load / shard / touch / scan it through
 * the ordinary entry points.  Owned by the communicator. */
pie_ctx *pie_comm_ctx(pie_comm *comm, int32_t rank);
/* Every local shard generates the synthetic corpus on its own GPU and keeps the rows of its users (pie_gen_synthetic +
 * pie_shard_table): the sharded form of BASELINE.json configs[3]. */
int pie_comm_gen_synthetic_sharded(pie_comm *comm, uint64_t seed, int64_t n_total, int32_t n_users, int32_t n_disc, uint32_t flags);
/* One synchronous step, per-query lists: every local shard runs ONE batched scan of the n_q queries and leaves its n_q result
 * messages ([off[0..u_pad] | M | rows], pie_pack_results_device's layout); the messages are exchanged (direct pattern: one
 * send and one receive per peer inside one ncclGroup); returns when every local rank holds all world x n_q messages.
 * u_pad: 0 in a single-process communicator (the largest shard's user count is used); in a process-per-GPU communicator
 * the value every rank agreed on, with the row capacity reserved beforehand (pie_comm_reserve) — the message length must be
 * the same everywhere.  m_out (may be NULL): local_ranks x n_q selected-row counts.
 * The exchange always runs (a list longer than the capacity is truncated, M stays in its header); the capacity check is made
 * on the gathered headers, so EVERY rank returns PIE_E_CAPACITY together: reserve pie_comm_needed_cap() and repeat. */
int pie_comm_scan_batch_gather(pie_comm *comm, const pie_query *queries, int32_t n_q, int32_t u_pad, size_t *m_out);
int pie_comm_reserve(pie_comm *comm, int32_t n_q, int32_t u_pad, size_t idx_cap);
/* Rows to reserve after PIE_E_CAPACITY (the largest list / union any rank reported, with headroom): the same on every rank. */
size_t pie_comm_needed_cap(const pie_comm *comm);
/* The gathered messages as rank `at_rank` holds them: device pointer and strides (words), or one message copied to the host. */
int pie_comm_gathered_device_ptr(pie_comm *comm, int32_t at_rank, void **base_out, size_t *rank_stride_words,
                                 size_t *query_stride_words, size_t *u_pad_out);
int pie_comm_read_gathered(pie_comm *comm, int32_t at_rank, int32_t src_rank, int32_t qi, int32_t *offsets_out /* u_pad + 1 */,
                           int32_t *idx_out, size_t idx_cap, size_t *m_out);

/* ---- the pipelined exchange: ONE union message per step (layout: pie_scan_batch_begin_union), written by each shard's own
 * batch kernels; the exchange of step i runs on a side stream while the shards scan steps i+1, i+2.  Order of calls:
 *     step_reserve;  begin(0); begin(1); finish(0); begin(2); collect(0); finish(1); begin(3); collect(1); ...
 * at most three steps per batch lane of the shards (pie_set_batch_lanes: 3 .. 12) begun and unfinished, at most sixteen
 * uncollected (rotating buffer sets).  Every rank of a process-per-GPU communicator makes the same calls in the same order.
 * step_reserve: u_pad as in pie_comm_scan_batch_gather; union_cap = rows per message.
 * step_finish:  waits for the oldest begun step's summaries on every local shard (m_out: local_ranks x n_q, may be NULL),
 *               then queues its exchange; does not wait for it.
 * step_collect: waits for the oldest queued exchange (side stream only); *step_out = its step number (0, 1, 2, ...).
 *               PIE_E_CAPACITY — on every rank alike, from the gathered Mu words — when a union outgrew union_cap (reserve
 *               pie_comm_needed_cap() once nothing is in flight, repeat) or could not be formed at all (Mu = -1). */
int pie_comm_step_reserve(pie_comm *comm, int32_t n_q, int32_t u_pad, size_t union_cap);
int pie_comm_step_begin(pie_comm *comm, const pie_query *queries, int32_t n_q);
int pie_comm_step_finish(pie_comm *comm, size_t *m_out);
int pie_comm_step_collect(pie_comm *comm, int64_t *step_out);
/* The gathered union messages of a collected step as rank `at_rank` holds them (valid until four more steps have begun):
 * message of rank r at base + r * rank_stride_words; or one message copied to the host (masks_out: 64-bit mask per row). */
int pie_comm_step_gathered_ptr(pie_comm *comm, int32_t at_rank, int64_t step, void **base_out, size_t *rank_stride_words,
                               size_t *u_pad_out, size_t *cap_out);
int pie_comm_step_read_gathered(pie_comm *comm, int32_t at_rank, int32_t src_rank, int64_t step, int32_t *uoff_out /* u_pad + 1 */,
                                int32_t *rows_out, uint64_t *masks_out, size_t cap, size_t *mu_out);

#ifdef __cplusplus
}
#endif
#endif /* PIE_SCAN_H */
