// pie_kernels.h — device kernels of the session-scan -> per-user feed path, written for gfx950 (CDNA4):
// 64-lane wavefronts, columns streamed with 16-B per-lane loads, per-wave ballot + mbcnt prefix compaction staged in
// LDS, one pass over the table.  Integer only — no MFMA on this path.
//
// Pipeline of one scan (all on one stream):
//   K1  the table pass: predicate, histogram counts[user] (the returning atomic is the row's rank in its bucket),
//       selected rows -> direct[user * 16 + rank] for rank < 16, else staged into the block's private region
//         k_scan_keyed       few rows live: streams a 1- or 2-byte liveness key, one 16-B payload gather per candidate
//         k_scan_live_first  the same idea on the 8-byte `end` column (tables without the derived columns)
//         k_scan_compact     many rows live: streams the predicate columns, late user materialisation
//   K2  k_offsets          one launch: exclusive scan of counts -> offsets[U+1] (tile granules with agent-scope
//                          stores, tiles claimed by ticket), in its fused form also the order of every bucket of
//                          <= 16 rows and the exchange message; work lists for K4, summary to the host through mapped
//                          memory, and the zeroing of the next scan's histogram span
//   K3  k_scatter          staged records -> bkt[offsets[u] + rank]          (only if a bucket outgrew 16 rows)
//   K4  k_sort_tiny        buckets of <= 16 rows, one thread each             (unfused K2 only)
//       k_sort_small       17..512 rows: one wave each, bitonic in registers + cross-lane exchanges
//       k_sort_segments    513..4096 rows and 4096-row tiles of big buckets: one 1024-thread block each, LDS bitonic
//   K4c k_merge_pass       big buckets only: log2(n/4096) rank-merge passes
// Order inside a bucket is (start asc, row index asc): ORDER BY start_ts ASC
// (/root/reference/server/storage/sqlProvider.js:276) with the tie rule of SURVEY.md §8 a-D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pie {

typedef long long ll2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

struct alignas(16) SelRec {
    long long start;
    int idx;
    int user;
};

// one bucket slot: the sort key (start, idx); one 16-B store / load per record
struct alignas(16) BktRec {
    long long start;
    int idx;
    int pad;
};

struct alignas(16) Segment {
    long long pos; // first slot in the bucket arrays
    int len;
    int flags; // bit0: tile of a big bucket -> write sorted (start, idx) pairs back in place
};

struct Summary {
    unsigned long long m;       // selected rows
    unsigned int n_seg;         // entries in the segment list
    unsigned int n_big;         // entries in the big-bucket list
    unsigned int max_count;     // largest bucket
    unsigned int bad_rows;      // rows whose user id fell outside [0, U): never selected, reported
    unsigned long long q;       // expired-queue length (pie_expired_queue)
    unsigned long long live;    // rows with end > now seen by K1 (drives the choice of K1 variant for the next scan)
    unsigned int n_small;       // entries in the small-segment list (17..256 rows: one wave each)
    unsigned int pad;
    unsigned long long amb;     // keyed table pass: rows whose liveness key equalled the query's (full `end` compare needed);
                                // streaming pass: selected rows whose lane neighbour selected a row of the same user
    unsigned int n_hot;         // users whose bucket exceeded the hot threshold (candidates for the next scan's hot set)
    unsigned int n_over;        // listed buckets with staged records (outgrew the direct slots / hot user / no slots): K3 needed
    unsigned long long cand;    // keyed table pass: candidate rows (key >= key(now)), i.e. payload records gathered
    unsigned int chunk_max;     // keyed table pass: candidates in the densest chunk (one load per lane: 1024 / 512 rows)
    unsigned int pad2;
};

// K2's inter-block state, zeroed together with the histogram it belongs to
struct ScanCtl {
    unsigned int ticket; // next tile to claim
    unsigned int done;   // tiles finished
    unsigned int pad[2];
};

// what the host reads after K2: written by the last K2 block into mapped pinned memory, seq last
struct HostSummary {
    Summary s;
    unsigned long long seq;
};

// Per-scan row statistics of the table pass (rows live, rows with an ambiguous key).  Thousands of K1 blocks adding to ONE
// address serialise at ~11 ns each in the L2 (a 6 000-block grid then cannot finish in under 130 us, whatever it
// streams), so the blocks spread over 64 counters on separate 128-byte lines, right behind the scan's Summary; the
// kernel that publishes the summary adds them up.
struct alignas(128) StatSlot {
    unsigned long long live, amb, cand;
    unsigned long long chunk_max; // max over the blocks of this slot
    unsigned long long max_count; // largest per-user count the blocks of this slot saw (k_ord_emit / k_ord_union_emit; k_ord_publish folds it in)
    unsigned long long pad[11];
};
constexpr int kStatSlots = 64;
constexpr int kSummaryBytes = 128; // Summary, padded: the slots start here
static_assert(sizeof(Summary) <= kSummaryBytes, "Summary outgrew its padded slot");
__device__ __forceinline__ StatSlot* stat_slots(Summary* s) { return reinterpret_cast<StatSlot*>(reinterpret_cast<char*>(s) + kSummaryBytes); }
__device__ __forceinline__ void add_row_stats(Summary* s, int live, int amb, int bid = (int)blockIdx.x, int cand = 0, int chunk_max = 0)
{
    StatSlot* slot = stat_slots(s) + (bid & (kStatSlots - 1));
    if (live) atomicAdd(&slot->live, (unsigned long long)live);
    if (amb) atomicAdd(&slot->amb, (unsigned long long)amb);
    if (cand) atomicAdd(&slot->cand, (unsigned long long)cand);
    if (chunk_max) atomicMax(&slot->chunk_max, (unsigned long long)chunk_max);
}
// one wave: lane l reads slot l; every lane returns the totals
__device__ __forceinline__ void sum_row_stats(Summary* s, int lane, unsigned long long& live, unsigned long long& amb,
                                              unsigned long long* cand = nullptr, unsigned int* chunk_max = nullptr)
{
    StatSlot* slot = stat_slots(s) + (lane & (kStatSlots - 1));
    live = __hip_atomic_load(&slot->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    amb = __hip_atomic_load(&slot->amb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long cd = __hip_atomic_load(&slot->cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long cm = __hip_atomic_load(&slot->chunk_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_xor(live, o, 64);
        amb += __shfl_xor(amb, o, 64);
        cd += __shfl_xor(cd, o, 64);
        const unsigned long long other = __shfl_xor(cm, o, 64);
        cm = other > cm ? other : cm;
    }
    if (cand) *cand = cd;
    if (chunk_max) *chunk_max = (unsigned int)cm;
}

constexpr int kWave = 64;
constexpr int kK1Threads = 256;
constexpr int kK1Waves = kK1Threads / kWave;
constexpr int kUnitRows = 2 * kWave;            // one int64x2 load per lane
constexpr int kStage = 128;                     // per-wave LDS ring of selected records
constexpr int kTinyMax = 16;
constexpr int kSmallMax = 512;                  // buckets of 17..512 rows: sorted by ONE wave in LDS, no block barriers
constexpr int kSegMax = 4096;
constexpr int kScanTile = 2048;                 // counts per K2 block of the unfused form (256 threads x 8)

// ------------------------------------------------------------------------------------------------ helpers

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `b` in lanes below the caller
__device__ __forceinline__ int prefix_in_ballot(unsigned long long b)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}

// Histogram index of a user.  The per-user counters K1 adds to are NOT laid out in user order: 32 consecutive users
// would share one 128-byte line, and the atomics of popular neighbours (ids handed out together, a popularity-sorted id
// space) then serialise on that line as if they hit one address.  Transposed instead: user u -> (u mod 32) * T + u / 32,
// T = ceil(U / 32), so consecutive users sit T counters apart.  K2 reads the histogram through the same map and
// writes the per-user counts the callers see in plain user order.
__device__ __forceinline__ int hist_index(int u, int n_users)
{
    const int t = (n_users + 31) >> 5;
    return (u & 31) * t + (u >> 5);
}

// Row predicate of SURVEY.md §8 a-D:
//   live    end > now            (dead iff expiresAt <= now, /root/reference/server/sessionStore.js:30,69)
//   window  start >= cutoff      (/root/reference/server/storage/sqlProvider.js:284)
//   disc    bit(mask, disc), ids outside the table never match (findDiscipline -> null,
//           /root/reference/server/disciplineConfig.js:88-97)
__device__ __forceinline__ bool row_selected(long long s, long long e, int d, long long now, long long cutoff,
                                             unsigned long long mask)
{
    const bool disc_ok = ((unsigned)d < 64u) && ((mask >> (d & 63)) & 1ull);
    return (e > now) & (s >= cutoff) & disc_ok;
}

__device__ __forceinline__ bool key_less(long long sa, int ia, long long sb, int ib)
{
    return (sa < sb) | ((sa == sb) & (ia < ib));
}

// ------------------------------------------------------------------------------------------------ K0 generator

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Synthetic corpus of SURVEY.md §8d — integer only, so the CPU checker reproduces it bit for bit.
__global__ __launch_bounds__(256) void k_gen(unsigned long long seed, long long n_total, long long row0, long long n,
                                             int n_users, int n_disc, unsigned flags, long long* __restrict__ start,
                                             long long* __restrict__ end, int* __restrict__ user,
                                             int* __restrict__ disc)
{
    constexpr unsigned long long G = 0x9E3779B97F4A7C15ULL;
    constexpr long long T0 = 1700000000000LL, SPAN = 10368000000LL, TTL = 43200000LL, MIN_DUR = 900000LL;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const unsigned long long i = (unsigned long long)(row0 + k);
        const unsigned long long r0 = mix64(seed + (4 * i + 1) * G), r1 = mix64(seed + (4 * i + 2) * G);
        const unsigned long long r2 = mix64(seed + (4 * i + 3) * G), r3 = mix64(seed + (4 * i + 4) * G);
        long long s = T0 - (long long)__umul64hi(r2, (unsigned long long)SPAN);
        // flags bit 2: rows in order of creation, as a session store appends them (start ascending with the row index,
        // evenly spread over the span): the live rows then sit together at the end of the table
        if (flags & 4u) s = T0 - SPAN + 1 + (long long)(((unsigned __int128)i * (unsigned long long)SPAN) / (unsigned long long)n_total);
        long long dur = TTL;
        if (flags & 1u) dur = MIN_DUR + (long long)__umul64hi(r3, (unsigned long long)(TTL - MIN_DUR + 1));
        start[k] = s;
        end[k] = s + dur;
        user[k] = (flags & 2u) ? (int)((i * (unsigned long long)n_users) / (unsigned long long)n_total)
                               : (int)__umul64hi(r0, (unsigned long long)n_users);
        disc[k] = (int)__umul64hi(r1, (unsigned long long)n_disc);
    }
}

// skewed users: user = first k with r0 < cdf[k] (thresholds supplied by the host), r0 = the row's first stream output
__global__ __launch_bounds__(256) void k_gen_users_cdf(unsigned long long seed, long long row0, long long n, int n_users,
                                                       const unsigned long long* __restrict__ cdf, int* __restrict__ user)
{
    constexpr unsigned long long G = 0x9E3779B97F4A7C15ULL;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const unsigned long long i = (unsigned long long)(row0 + k);
        const unsigned long long r0 = mix64(seed + (4 * i + 1) * G);
        int lo = 0, hi = n_users - 1;
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (r0 < cdf[mid]) hi = mid; else lo = mid + 1;
        }
        user[k] = lo;
    }
}

// ------------------------------------------------------------------------------------------------ K1 scan + compact

struct WaveStage {
    SelRec* ring;  // this wave's LDS ring (kStage records)
    int* ring_rank;
    int head;      // wave-uniform
    int fill;      // wave-uniform, < 64 between calls
};

// Flush staged records: one LDS atomic per wave reserves slots of the block's private output range; every
// lane stores one 16-B record + its 4-B rank (1 KiB + 256 B coalesced per wave).
__device__ __forceinline__ void stage_flush(WaveStage& st, int count, SelRec* __restrict__ out, int* __restrict__ out_rank,
                                            int* blk_cursor, int lane)
{
    int base = 0;
    if (lane == 0) base = atomicAdd(blk_cursor, count);
    base = __builtin_amdgcn_readfirstlane(base);
    if (lane < count) {
        const int slot = (st.head + lane) & (kStage - 1);
        out[base + lane] = st.ring[slot];
        out_rank[base + lane] = st.ring_rank[slot];
    }
    st.head = (st.head + count) & (kStage - 1);
    st.fill -= count;
}

__device__ __forceinline__ void stage_rows(bool sel, long long s, int row, int u, int rank, WaveStage& st,
                                           SelRec* __restrict__ out, int* __restrict__ out_rank, int* blk_cursor, int lane)
{
    const unsigned long long b = __ballot(sel);
    if (b == 0) return; // wave-uniform: nothing selected in this 64-row slice
    if (sel) {
        const int slot = (st.head + st.fill + prefix_in_ballot(b)) & (kStage - 1);
        SelRec r;
        r.start = s;
        r.idx = row;
        r.user = u;
        st.ring[slot] = r;
        st.ring_rank[slot] = rank;
    }
    st.fill += __popcll(b);
    __builtin_amdgcn_wave_barrier();
    if (st.fill >= kWave) stage_flush(st, kWave, out, out_rank, blk_cursor, lane);
    __builtin_amdgcn_wave_barrier();
}

// Direct bucket slots.  A selected row whose rank inside its user's bucket (the value its histogram atomic returned) is
// below the slot capacity goes straight to direct[(user << shift) + rank]: a fixed-capacity bucket per user that needs no
// offsets, so nothing is staged for it and K3 has nothing to scatter.  The capacity is 16 (the size the fused K2 orders
// in registers) until a scan finds buckets that outgrow it; the host then grows it (powers of two up to kSmallMax, the
// size one wave orders) so that dense queries skip the staging + scatter pass as well.  Only ranks >= capacity take the
// staged route.  p == nullptr (user tables too large to carry the slots) keeps every row on the staged route.
struct DirectSlots {
    BktRec* p;
    int shift; // capacity per user = 1 << shift
};
__device__ __forceinline__ BktRec* slots_of(const DirectSlots& d, int u) { return d.p + ((long long)u << d.shift); }
// Hot users.  A user who owns a large share of the selected rows (a Zipf head) turns the histogram into same-address
// atomics, which serialise at ~11 ns each however they are aggregated per wave.  For up to kHotMax such users (the big
// buckets of the previous scan; the ids travel as kernel arguments, i.e. in SGPRs) the wave-aggregated forms count
// per BLOCK in LDS instead: a hot row takes its rank inside the block from an LDS counter and is staged with the rank
// flagged (bit 31), the block adds its per-user totals to counts[] once at the end and records the bases it got, and
// K3 places the record at offsets[user] + base(block, user) + rank-in-block.  Hot rows never use the direct slots.
constexpr int kHotMax = 32;
constexpr int kHotFlag = (int)0x80000000;
struct HotSet {
    int n;
    int user[kHotMax];
};
__device__ __forceinline__ int hot_slot_of(const HotSet& hot, int u)
{
    int slot = -1;
#pragma unroll
    for (int k = 0; k < kHotMax; ++k)
        if (k < hot.n && u == hot.user[k]) slot = k;
    return slot;
}

__device__ __forceinline__ void emit_row(bool sel, long long s, int row, int u, int rank, const DirectSlots& direct,
                                         WaveStage& st, SelRec* __restrict__ out, int* __restrict__ out_rank, int* blk_cursor,
                                         int lane)
{
    if (direct.p && sel && (unsigned)rank < (1u << direct.shift)) { // flagged (hot) ranks are negative: never direct
        BktRec r;
        r.start = s;
        r.idx = row;
        r.pad = 0;
        slots_of(direct, u)[rank] = r;
        sel = false;
    }
    stage_rows(sel, s, row, u, rank, st, out, out_rank, blk_cursor, lane);
}

template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T* p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// Block b owns rows [b*rows_per_block, (b+1)*rows_per_block) and the same index range of `sel` as its
// private output region (a block can never select more rows than it reads), so compaction needs no global
// cursor: blk_count[b] says how many records the region holds.
//   UNROLL  128-row units per wave iteration
//   NT      nontemporal (streaming) loads for the read-once columns
//   LATE_U  load user[] only in lanes whose row passed the predicate (late materialisation): the user column
//           is output data, not predicate input, so unselected rows never need it
//   GQ      group-qualified form (archive queue): the row predicate is "row not tombstoned and qual[user] != 0" and the
//           staged sort key is 0, so the per-bucket order is plain row order
//   AGG     wave-aggregated histogram atomics: in each 64-row slice the lanes of one user form a group and issue ONE
//           returning atomic.  For tables whose rows are clustered by user (the "best case" order) a dense query
//           otherwise hammers one counter from all 64 lanes; chosen by the host from the share of selected rows whose
//           lane neighbour has the same user, which every non-aggregated pass of this kernel counts.
template <int UNROLL, bool NT, bool LATE_U, bool GQ = false, bool AGG = false>
__device__ __forceinline__ void scan_compact_body(
    const long long* __restrict__ start, const long long* __restrict__ end, const int* __restrict__ user,
    const int* __restrict__ disc, long long n, long long rows_per_block, long long now, long long cutoff,
    unsigned long long mask, int n_users, int* __restrict__ counts, SelRec* __restrict__ sel,
    int* __restrict__ sel_rank, int* __restrict__ blk_count, Summary* __restrict__ summary, DirectSlots direct,
    const unsigned char* __restrict__ qual, const int bid)
{
    static_assert(!(GQ && LATE_U), "the group-qualified predicate needs the user column up front");
    __shared__ SelRec stage[kK1Waves][kStage];
    __shared__ int stage_rank[kK1Waves][kStage];
    __shared__ int blk_cursor;
    __shared__ int blk_live;
    __shared__ int blk_dup;
    constexpr int kTile = kUnitRows * UNROLL;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    unsigned int* bad_rows = &summary->bad_rows;
    int nlive = 0; // wave-uniform count of rows with end > now
    int ndup = 0;  // wave-uniform count of selected rows whose nearest selected lane below holds the same user
    if (threadIdx.x == 0) { blk_cursor = 0; blk_live = 0; blk_dup = 0; }
    __syncthreads();

    const long long c0 = (long long)bid * rows_per_block;
    long long c1 = c0 + rows_per_block;
    if (c1 > n) c1 = n;
    SelRec* out = sel + c0;
    int* out_rank = sel_rank + c0;
    WaveStage st;
    st.ring = stage[wave];
    st.ring_rank = stage_rank[wave];
    st.head = 0;
    st.fill = 0;

    for (long long t = c0 + (long long)wave * kTile; t < c1; t += (long long)kTile * kK1Waves) {
        if (t + kTile <= c1) {
            // full wave-tile: UNROLL units x (2 rows per lane); every load is a fully used, aligned line
            ll2_t s[UNROLL], e[UNROLL];
            i2_t u[UNROLL], d[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const long long r = t + j * kUnitRows + 2 * lane;
                e[j] = stream_load<NT>(reinterpret_cast<const ll2_t*>(end + r));
                s[j] = stream_load<NT>(reinterpret_cast<const ll2_t*>(start + r));
                d[j] = stream_load<NT>(reinterpret_cast<const i2_t*>(disc + r));
                if constexpr (!LATE_U) u[j] = stream_load<NT>(reinterpret_cast<const i2_t*>(user + r));
            }
            bool p[2 * UNROLL];
            int rank[2 * UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                if constexpr (GQ) {
                    p[2 * j] = (e[j].x != INT64_MIN) && (unsigned)u[j].x < (unsigned)n_users && qual[u[j].x];
                    p[2 * j + 1] = (e[j].y != INT64_MIN) && (unsigned)u[j].y < (unsigned)n_users && qual[u[j].y];
                    s[j].x = 0;
                    s[j].y = 0;
                } else {
                    p[2 * j] = row_selected(s[j].x, e[j].x, d[j].x, now, cutoff, mask);
                    p[2 * j + 1] = row_selected(s[j].y, e[j].y, d[j].y, now, cutoff, mask);
                }
                nlive += __popcll(__ballot(e[j].x > now)) + __popcll(__ballot(e[j].y > now));
                if constexpr (LATE_U) {
                    const long long r = t + j * kUnitRows + 2 * lane;
                    u[j].x = 0;
                    u[j].y = 0;
                    if (p[2 * j]) u[j].x = user[r];
                    if (p[2 * j + 1]) u[j].y = user[r + 1];
                }
            }
            // phase A: histogram + rank.  The returning atomic's latency hides behind the streaming loads of
            // the other waves; K3 then needs no atomics at all.
#pragma unroll
            for (int k = 0; k < 2 * UNROLL; ++k) {
                const int uu = (k & 1) ? u[k >> 1].y : u[k >> 1].x;
                rank[k] = 0;
                if (p[k] && (unsigned)uu >= (unsigned)n_users) { atomicAdd(bad_rows, 1u); p[k] = false; }
                if constexpr (AGG) {
                    int grp_leader = lane, grp_prefix = 0, grp_size = 1;
                    unsigned long long todo = __ballot(p[k]);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int u_lead = __shfl(uu, leader, kWave);
                        const unsigned long long same = __ballot(p[k] && uu == u_lead);
                        if (p[k] && uu == u_lead) {
                            grp_leader = leader;
                            grp_prefix = prefix_in_ballot(same);
                            grp_size = __popcll(same);
                        }
                        todo &= ~same;
                    }
                    int base = 0;
                    if (p[k] && grp_leader == lane) base = atomicAdd(&counts[hist_index(uu, n_users)], grp_size);
                    rank[k] = __shfl(base, grp_leader, kWave) + grp_prefix;
                    // reported instead of the neighbour count: selected rows that did NOT need an atomic of their own
                    ndup += __popcll(__ballot(p[k] && grp_leader != lane));
                } else {
                    if (p[k]) rank[k] = atomicAdd(&counts[hist_index(uu, n_users)], 1);
                    // does the nearest selected lane below this one hold the same user?
                    const unsigned long long below = __ballot(p[k]) & ((1ull << lane) - 1ull);
                    const int prev = below ? 63 - __clzll((long long)below) : 0;
                    const int u_prev = __shfl(uu, prev, kWave);
                    ndup += __popcll(__ballot(p[k] && below != 0 && uu == u_prev));
                }
            }
            // phase B: wave-prefix compaction into the LDS ring
#pragma unroll
            for (int k = 0; k < 2 * UNROLL; ++k) {
                const int j = k >> 1;
                const int r = (int)(t + j * kUnitRows + 2 * lane) + (k & 1);
                emit_row(p[k], (k & 1) ? s[j].y : s[j].x, r, (k & 1) ? u[j].y : u[j].x, rank[k], direct, st, out, out_rank,
                         &blk_cursor, lane);
            }
        } else {
            // ragged tail of the block's range: one row per lane, bounds-checked
            const long long t1 = (t + kTile < c1) ? t + kTile : c1;
            for (long long r0 = t; r0 < t1; r0 += kWave) {
                const long long r = r0 + lane;
                bool sel_row = false;
                long long sv = 0;
                int uv = 0, rk = 0;
                const long long ev = r < t1 ? end[r] : INT64_MIN;
                nlive += __popcll(__ballot(ev > now));
                if (r < t1) {
                    sv = start[r];
                    sel_row = row_selected(sv, ev, disc[r], now, cutoff, mask);
                    if constexpr (GQ) {
                        const int ug = user[r];
                        sel_row = (ev != INT64_MIN) && (unsigned)ug < (unsigned)n_users && qual[ug];
                        sv = 0;
                    }
                    if (sel_row) {
                        uv = user[r];
                        if ((unsigned)uv < (unsigned)n_users) rk = atomicAdd(&counts[hist_index(uv, n_users)], 1);
                        else { atomicAdd(bad_rows, 1u); sel_row = false; }
                    }
                }
                emit_row(sel_row, sv, (int)r, uv, rk, direct, st, out, out_rank, &blk_cursor, lane);
            }
        }
    }
    if (st.fill > 0) stage_flush(st, st.fill, out, out_rank, &blk_cursor, lane);
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    if (lane == 0 && ndup) atomicAdd(&blk_dup, ndup);
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_count[bid] = blk_cursor;
        // the streaming form has no ambiguous keys: the second statistic carries its same-user-neighbour count instead
        add_row_stats(summary, blk_live, blk_dup, bid);
    }
}

template <int UNROLL, bool NT, bool LATE_U, bool GQ = false, bool AGG = false>
__global__ __launch_bounds__(kK1Threads) void k_scan_compact(
    const long long* __restrict__ start, const long long* __restrict__ end, const int* __restrict__ user,
    const int* __restrict__ disc, long long n, long long rows_per_block, long long now, long long cutoff,
    unsigned long long mask, int n_users, int* __restrict__ counts, SelRec* __restrict__ sel,
    int* __restrict__ sel_rank, int* __restrict__ blk_count, Summary* __restrict__ summary, DirectSlots direct,
    const unsigned char* __restrict__ qual = nullptr)
{
    scan_compact_body<UNROLL, NT, LATE_U, GQ, AGG>(start, end, user, disc, n, rows_per_block, now, cutoff, mask, n_users, counts, sel, sel_rank,
                                                   blk_count, summary, direct, qual, (int)blockIdx.x);
}

// K1, liveness-first form.  In a session store almost every row is expired (12 h TTL, months of history), so
// `end > now` is by far the most selective conjunct.  This form streams ONLY the `end` column (8 B/row), keeps
// the indices of live rows in a per-wave LDS ring, and whenever 64 of them are queued evaluates the rest of the
// predicate for those rows with all 64 lanes busy: start/disc/user are gathered for live rows only (late
// materialisation of three columns).  Output is identical to k_scan_compact; the host picks the form from the
// live fraction the previous scan observed.
constexpr int kLiveRing = 128;

template <int UNROLL, bool NT, bool AGG>
__global__ __launch_bounds__(kK1Threads) void k_scan_live_first(
    const long long* __restrict__ start, const long long* __restrict__ end, const int* __restrict__ user,
    const int* __restrict__ disc, long long n, long long rows_per_block, long long now, long long cutoff,
    unsigned long long mask, int n_users, int* __restrict__ counts, SelRec* __restrict__ sel,
    int* __restrict__ sel_rank, int* __restrict__ blk_count, Summary* __restrict__ summary, DirectSlots direct,
    HotSet hot, int* __restrict__ blk_hot_base)
{
    __shared__ SelRec stage[kK1Waves][kStage];
    __shared__ int stage_rank[kK1Waves][kStage];
    __shared__ int live_ring[kK1Waves][kLiveRing];
    __shared__ int blk_cursor;
    __shared__ int blk_live;
    __shared__ int blk_hot_cnt[kHotMax];
    constexpr int kTile = kUnitRows * UNROLL;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_cursor = 0; blk_live = 0; }
    if (AGG && threadIdx.x < kHotMax) blk_hot_cnt[threadIdx.x] = 0;
    __syncthreads();

    const long long c0 = (long long)blockIdx.x * rows_per_block;
    long long c1 = c0 + rows_per_block;
    if (c1 > n) c1 = n;
    SelRec* out = sel + c0;
    int* out_rank = sel_rank + c0;
    WaveStage st;
    st.ring = stage[wave];
    st.ring_rank = stage_rank[wave];
    st.head = 0;
    st.fill = 0;
    int* lring = live_ring[wave];
    int lhead = 0, lfill = 0, nlive = 0; // wave-uniform

    // evaluate the rest of the predicate for `cnt` queued live rows (cnt <= 64), one row per lane
    auto drain = [&](int cnt) {
        bool p = false;
        long long sv = 0;
        int row = 0, uv = -1, rk = 0;
        if (lane < cnt) {
            row = lring[(lhead + lane) & (kLiveRing - 1)];
            sv = start[row];
            const int dv = disc[row];
            p = (sv >= cutoff) & ((unsigned)dv < 64u) & (((mask >> (dv & 63)) & 1ull) != 0);
            if (p) {
                uv = user[row];
                if ((unsigned)uv >= (unsigned)n_users) { atomicAdd(&summary->bad_rows, 1u); p = false; }
            }
        }
        // Histogram + rank with wave-level aggregation: lanes of one user form a group, the group's first lane adds
        // the group size with ONE returning atomic and the members take consecutive ranks.  A skewed table (Zipf
        // users) otherwise hammers a few addresses with same-address atomics, which serialise (K1 4.5x slower).
        // The grouping loop is pure ALU (one pass per distinct user in the batch) and runs once per 64 LIVE rows.
        // AGG is chosen by the host when the previous scan saw one bucket holding > 1/64 of the selected rows.
        // hot users first: rank inside the block from an LDS counter, flagged; they take no part in the grouping below
        bool hotrow = false;
        int hot_rank = 0;
        if constexpr (AGG) {
            const int hs = p ? hot_slot_of(hot, uv) : -1;
            hotrow = hs >= 0;
            if (hotrow) hot_rank = atomicAdd(&blk_hot_cnt[hs], 1) | kHotFlag;
        }
        const bool pg = p && !hotrow;
        int grp_leader = lane, grp_prefix = 0, grp_size = 1;
        unsigned long long todo = AGG ? __ballot(pg) : 0ull;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int u_lead = __shfl(uv, leader, kWave);
            const unsigned long long same = __ballot(pg && uv == u_lead);
            if (pg && uv == u_lead) {
                grp_leader = leader;
                grp_prefix = prefix_in_ballot(same);
                grp_size = __popcll(same);
            }
            todo &= ~same;
        }
        int base = 0;
        if (pg && grp_leader == lane) base = atomicAdd(&counts[hist_index(uv, n_users)], grp_size);
        base = __shfl(base, grp_leader, kWave);
        rk = hotrow ? hot_rank : base + grp_prefix;
        lhead = (lhead + cnt) & (kLiveRing - 1);
        lfill -= cnt;
        emit_row(p, sv, row, uv, rk, direct, st, out, out_rank, &blk_cursor, lane);
    };
    auto push_live = [&](bool live, int row) {
        const unsigned long long b = __ballot(live);
        if (b == 0) return;
        if (live) lring[(lhead + lfill + prefix_in_ballot(b)) & (kLiveRing - 1)] = row;
        const int c = __popcll(b);
        lfill += c;
        nlive += c;
        __builtin_amdgcn_wave_barrier();
        if (lfill >= kWave) drain(kWave);
        __builtin_amdgcn_wave_barrier();
    };

    for (long long t = c0 + (long long)wave * kTile; t < c1; t += (long long)kTile * kK1Waves) {
        if (t + kTile <= c1) {
            ll2_t e[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j)
                e[j] = stream_load<NT>(reinterpret_cast<const ll2_t*>(end + t + j * kUnitRows + 2 * lane));
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const int r = (int)(t + j * kUnitRows + 2 * lane);
                push_live(e[j].x > now, r);
                push_live(e[j].y > now, r + 1);
            }
        } else {
            const long long t1 = (t + kTile < c1) ? t + kTile : c1;
            for (long long r0 = t; r0 < t1; r0 += kWave) {
                const long long r = r0 + lane;
                push_live(r < t1 && end[r] > now, (int)r);
            }
        }
    }
    if (lfill > 0) drain(lfill);
    if (st.fill > 0) stage_flush(st, st.fill, out, out_rank, &blk_cursor, lane);
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_count[blockIdx.x] = blk_cursor;
        add_row_stats(summary, blk_live, 0);
    }
    if constexpr (AGG) { // one histogram atomic per (block, hot user); K3 needs the base it returned
        if ((int)threadIdx.x < hot.n) {
            const int cnt = blk_hot_cnt[threadIdx.x];
            blk_hot_base[(long long)blockIdx.x * kHotMax + threadIdx.x] = cnt ? atomicAdd(&counts[hist_index(hot.user[threadIdx.x], n_users)], cnt) : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ liveness-key column
//
// The resident table carries one derived column next to `end`: a 15-bit monotone key of it in a uint16,
//     key(e) = 0                                    e <  base          (tombstones, INT64_MIN, land here)
//            = min(((e - base) >> shift) + 1, 32767) e >= base
// built at load time with base = the smallest live `end` and the smallest shift that keeps the largest one below the
// clamp.  key() is monotone non-decreasing over all of int64, so for any query time
//     key(end) > key(now)  =>  end > now      key(end) < key(now)  =>  end < now
// and only rows with key(end) == key(now) need the full 8-byte compare.  The liveness-first table pass then streams
// 2 B/row instead of 8 B/row; correctness never depends on how well base/shift fit the data (a bad fit only makes more
// rows ambiguous, which the scan reports in Summary::amb so the host can rebuild the column).
// Every writer of `end` (k_set_end, the tombstoning list kernels, load / append / generate) keeps the key in step.
//
// A second, finer-grained key covers only the top of the range: the liveness-first forms are chosen when fewer than a
// tenth of the rows are live, i.e. when `now` lies above the 90th percentile of `end`, so a 7-bit key in a uint8 whose
// base sits at that percentile (found from a histogram of the 15-bit keys) separates every such query with 1 B/row.
// Same key function, same exactness argument, other constants; queries below its base use the 15-bit key.
typedef unsigned short lkey_t;
typedef unsigned char fkey_t;
constexpr unsigned kKeyMax = 32767u;
constexpr unsigned kFineKeyMax = 127u;
constexpr int kKeyHistBins = 4096; // histogram of (15-bit key >> 3)

__device__ __forceinline__ unsigned key_of(long long e, long long base, int shift, unsigned kmax = kKeyMax)
{
    if (e < base) return 0u;
    const unsigned long long k = ((unsigned long long)e - (unsigned long long)base) >> shift;
    return k >= (unsigned long long)(kmax - 1u) ? kmax : (unsigned)k + 1u;
}

// The ordered run (pie_ordered.h) keeps its own copies of `end` and of both keys, in (user, start, row) order.  Every writer of
// `end` that does not invalidate the run mirrors its store through the row -> position map; a row the run does not hold
// (tombstoned when the run was built) that comes back to life is counted in *stale (mapped host memory): the host call
// that made the store reads it when its kernel has finished and drops the run.
struct OrdMirror {
    const int* pos;        // nullptr: no ordered run
    long long* end;
    lkey_t* key;
    fkey_t* fkey;
    unsigned int* stale;
};
__device__ __forceinline__ void ord_mirror_end(const OrdMirror& o, long long row, long long e, unsigned k, unsigned fk)
{
    if (!o.pos) return;
    const int p = o.pos[row];
    if (p >= 0) {
        o.end[p] = e;
        o.key[p] = (lkey_t)k;
        o.fkey[p] = (fkey_t)fk;
    } else if (e != INT64_MIN) __hip_atomic_fetch_add(o.stale, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // mapped host memory
}

// histogram of the 15-bit keys of rows [0, n), 8 keys per bin, privatised in LDS
__global__ __launch_bounds__(1024) void k_key_hist(const lkey_t* __restrict__ key, long long n, unsigned int* __restrict__ hist)
{
    __shared__ unsigned int h[kKeyHistBins];
    for (int i = threadIdx.x; i < kKeyHistBins; i += 1024) h[i] = 0;
    __syncthreads();
    for (long long r = (long long)blockIdx.x * 1024 + threadIdx.x; r < n; r += (long long)gridDim.x * 1024) atomicAdd(&h[key[r] >> 3], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < kKeyHistBins; i += 1024)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// fine keys of rows [row0, n)
__global__ __launch_bounds__(256) void k_build_fine_key(const long long* __restrict__ end, long long row0, long long n, long long base,
                                                        int shift, fkey_t* __restrict__ fkey)
{
    for (long long r = row0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x)
        fkey[r] = (fkey_t)key_of(end[r], base, shift, kFineKeyMax);
}

// smallest and largest `end` of the rows that are not tombstoned (range[0] = INT64_MAX, range[1] = INT64_MIN going in)
__global__ __launch_bounds__(256) void k_end_range(const long long* __restrict__ end, long long n, long long* __restrict__ range)
{
    long long lo = INT64_MAX, hi = INT64_MIN;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const long long e = end[r];
        if (e != INT64_MIN) {
            lo = e < lo ? e : lo;
            hi = e > hi ? e : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long l2 = __shfl_xor(lo, o, kWave), h2 = __shfl_xor(hi, o, kWave);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&range[0], lo);
        atomicMax(&range[1], hi);
    }
}

// The second derived column: the three immutable fields of a row side by side, so that a candidate row costs ONE
// 16-byte gather (one 128-byte HBM sector) instead of three gathers in three columns.  Written once per row (load,
// append, generate); nothing ever changes start / user / disc afterwards.
struct alignas(16) PayRec {
    long long start;
    int user;
    int disc;
};

__global__ __launch_bounds__(256) void k_build_key(const long long* __restrict__ end, long long row0, long long n, long long base,
                                                   int shift, lkey_t* __restrict__ key, const long long* __restrict__ start,
                                                   const int* __restrict__ user, const int* __restrict__ disc,
                                                   PayRec* __restrict__ pay)
{
    for (long long r = row0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        key[r] = (lkey_t)key_of(end[r], base, shift);
        if (pay) {
            PayRec p;
            p.start = start[r];
            p.user = user[r];
            p.disc = disc[r];
            pay[r] = p;
        }
    }
}

// K1, keyed liveness-first form: streams the 2-byte key column (eight rows per 16-byte load), finds the candidate rows
// (key >= key(now)) with a SWAR compare, and from there on works like k_scan_live_first: candidates queue up in a
// per-wave LDS ring and are evaluated 64 at a time — `end` only for the ambiguous ones, and one 16-byte record of the
// payload column (start, user, disc) per candidate.  Output identical to the other forms.
// KT = lkey_t: the 15-bit key, 8 rows per 16-byte load; KT = fkey_t: the 7-bit top-of-range key, 16 rows per load.
constexpr int kKeyRowsPerLoad = 8 * kWave;      // 512 rows per wave per 16-byte load (2-byte keys)
constexpr int kFineKeyRowsPerLoad = 16 * kWave; // 1024 rows (1-byte keys)

// PIPE: the evaluation of a batch of 64 candidates is split into three steps that run one batch apart — (A) take the batch
// off the ring and issue its gathers, (B) one batch later, evaluate the predicate and issue the histogram atomics,
// (C) one batch later again, take the returned ranks and emit — so neither the gather's nor the atomic's round trip
// stalls the wave: the key stream keeps flowing in between.
template <int UNROLL, bool AGG, bool NT, class KT, bool PIPE>
__device__ __forceinline__ void scan_keyed_body(
    const PayRec* __restrict__ pay, const long long* __restrict__ end, const KT* __restrict__ key, long long n,
    long long rows_per_block, long long now, unsigned now_key, long long cutoff, unsigned long long mask, int n_users, int* __restrict__ counts,
    SelRec* __restrict__ sel, int* __restrict__ sel_rank, int* __restrict__ blk_count, Summary* __restrict__ summary,
    DirectSlots direct, const HotSet& hot, int* __restrict__ blk_hot_base, int bid, int n_scan_blocks, int run_shift_arg)
{
    __shared__ SelRec stage[kK1Waves][kStage];
    __shared__ int stage_rank[kK1Waves][kStage];
    __shared__ int live_ring[kK1Waves][kLiveRing];
    __shared__ int blk_cursor;
    __shared__ int blk_live;
    __shared__ int blk_amb;
    __shared__ int blk_cand;
    __shared__ int blk_chunk_max;
    __shared__ int blk_hot_cnt[kHotMax];
    constexpr int kPerLane = 16 / (int)sizeof(KT); // rows per lane per 16-byte load
    constexpr int kLogUnroll = UNROLL >= 8 ? 3 : UNROLL >= 4 ? 2 : UNROLL >= 2 ? 1 : 0;
    const int run_shift = run_shift_arg < kLogUnroll ? (run_shift_arg < 0 ? 0 : run_shift_arg) : kLogUnroll;
    int pushed = 0, chunk_max = 0; // wave-uniform: candidates queued so far; most candidates seen in one chunk
    constexpr int kRowsPerLoad = kPerLane * kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_cursor = 0; blk_live = 0; blk_amb = 0; blk_cand = 0; blk_chunk_max = 0; }
    if (AGG && threadIdx.x < kHotMax) blk_hot_cnt[threadIdx.x] = 0;
    __syncthreads();

    // the block's private output region for staged records: rows_per_block slots (it never reads more rows than that)
    const long long c0 = (long long)bid * rows_per_block;
    SelRec* out = sel + c0;
    int* out_rank = sel_rank + c0;
    WaveStage st;
    st.ring = stage[wave];
    st.ring_rank = stage_rank[wave];
    st.head = 0;
    st.fill = 0;
    int* lring = live_ring[wave];
    int lhead = 0, lfill = 0, nlive = 0, namb = 0, ncand = 0; // wave-uniform

    // evaluate `cnt` queued candidates (cnt <= 64), one per lane; bit 31 of a ring entry marks an ambiguous key
    // step A state (gathers in flight), step B state (atomics in flight); all "have" flags are wave-uniform
    bool a_have = false, a_valid = false, a_amb = false;
    int a_row = 0;
    PayRec a_pay;
    a_pay.start = 0; a_pay.user = 0; a_pay.disc = 0;
    long long a_end = 0;
    bool b_have = false, b_p = false;
    long long b_sv = 0;
    int b_row = 0, b_uv = -1, b_base = 0, b_leader = 0, b_prefix = 0;

    auto step_c = [&]() { // ranks are back: emit
        if (!b_have) return;
        const int base = __shfl(b_base, b_leader, kWave);
        emit_row(b_p, b_sv, b_row, b_uv, base + b_prefix, direct, st, out, out_rank, &blk_cursor, lane);
        b_have = false;
    };
    auto step_b = [&]() { // gathers are back: predicate, histogram atomics
        if (!a_have) return;
        bool p = false, live = false;
        long long sv = 0;
        int uv = -1;
        if (a_valid) {
            live = a_amb ? (a_end > now) : true;
            sv = a_pay.start;
            const int dv = a_pay.disc;
            p = live & (sv >= cutoff) & ((unsigned)dv < 64u) & (((mask >> (dv & 63)) & 1ull) != 0);
            if (p) {
                uv = a_pay.user;
                if ((unsigned)uv >= (unsigned)n_users) { atomicAdd(&summary->bad_rows, 1u); p = false; }
            }
        }
        nlive += __popcll(__ballot(live));
        namb += __popcll(__ballot(a_valid && a_amb));
        ncand += __popcll(__ballot(a_valid));
        // hot users: rank inside the block from an LDS counter, flagged (see HotSet); the rest: wave-aggregated
        // histogram atomics for skewed users (see k_scan_live_first)
        bool hotrow = false;
        int hot_rank = 0;
        if constexpr (AGG) {
            const int hs = p ? hot_slot_of(hot, uv) : -1;
            hotrow = hs >= 0;
            if (hotrow) hot_rank = atomicAdd(&blk_hot_cnt[hs], 1) | kHotFlag;
        }
        const bool pg = p && !hotrow;
        int grp_leader = lane, grp_prefix = 0, grp_size = 1;
        unsigned long long todo = AGG ? __ballot(pg) : 0ull;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int u_lead = __shfl(uv, leader, kWave);
            const unsigned long long same = __ballot(pg && uv == u_lead);
            if (pg && uv == u_lead) {
                grp_leader = leader;
                grp_prefix = prefix_in_ballot(same);
                grp_size = __popcll(same);
            }
            todo &= ~same;
        }
        int base = 0;
        if (pg && grp_leader == lane) base = atomicAdd(&counts[hist_index(uv, n_users)], grp_size);
        if (hotrow) { base = hot_rank; grp_leader = lane; grp_prefix = 0; } // step C adds nothing to a flagged rank
        b_have = true;
        b_p = p;
        b_sv = sv;
        b_row = a_row;
        b_uv = uv;
        b_base = base;
        b_leader = grp_leader;
        b_prefix = grp_prefix;
        a_have = false;
    };
    auto step_a = [&](int cnt) { // take the batch off the ring, issue its gathers
        a_valid = lane < cnt;
        a_amb = false;
        if (a_valid) {
            const int ent = lring[(lhead + lane) & (kLiveRing - 1)];
            a_row = ent & 0x7FFFFFFF;
            a_amb = ent < 0;
            a_pay = pay[a_row]; // one gather per candidate
            if (a_amb) a_end = end[a_row];
        }
        a_have = true;
        lhead = (lhead + cnt) & (kLiveRing - 1);
        lfill -= cnt;
    };
    auto drain = [&](int cnt) {
        if constexpr (PIPE) {
            step_c();
            step_b();
            step_a(cnt);
        } else {
            step_a(cnt);
            step_b();
            step_c();
        }
    };
    auto push = [&](bool cand, int entry) {
        const unsigned long long b = __ballot(cand);
        if (b == 0) return false;
        if (cand) lring[(lhead + lfill + prefix_in_ballot(b)) & (kLiveRing - 1)] = entry;
        lfill += __popcll(b);
        pushed += __popcll(b);
        __builtin_amdgcn_wave_barrier();
        if (lfill >= kWave) drain(kWave);
        __builtin_amdgcn_wave_barrier();
        return true;
    };

    // SWAR: keys are < 2^15 (< 2^7), so with the top bit of each half (byte) forced on, subtracting key(now) from all of
    // them at once never borrows across, and the top bit of a half (byte) survives exactly when that key >= key(now)
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr unsigned kTop = sizeof(KT) == 2 ? 0x80008000u : 0x80808080u;
    const unsigned nkr = sizeof(KT) == 2 ? (now_key | (now_key << 16)) : now_key * 0x01010101u;
    // Rows are dealt to the waves of the whole launch in CHUNKS of one 16-byte load per lane (1024 / 512 rows), round robin:
    // wave g of W takes chunks g, g + W, g + 2W, ...  In a session table the live rows are the recent ones, i.e. they sit
    // together at the end of the table (and a login burst appends there): with one contiguous row range per block the few
    // blocks at the end would evaluate nearly every candidate while the rest of the chip idles (measured: the table pass
    // 4x slower once 10^5 freshly appended / touched rows are live).  Interleaved, a dense stretch of the table is spread
    // over as many waves as it has chunks.  Every load is still one contiguous, aligned KiB per wave.
    int chunk_mark = 0;
    const long long n_chunks = n / kRowsPerLoad;              // full chunks; the ragged rest goes to one wave, row by row
    const long long W = (long long)n_scan_blocks * kK1Waves;
    const long long gw = (long long)bid * kK1Waves + wave;
    for (long long cb = gw << run_shift; cb < n_chunks; cb += W * UNROLL) {
        u4_t kv[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const long long ch = cb + (((long long)(j >> run_shift) * W) << run_shift) + (j & ((1 << run_shift) - 1));
            kv[j] = (u4_t){0u, 0u, 0u, 0u};                    // key 0 = below every query's key: no candidates
            if (ch < n_chunks) kv[j] = stream_load<NT>(reinterpret_cast<const u4_t*>(key + ch * kRowsPerLoad + kPerLane * lane));
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const long long ch = cb + (((long long)(j >> run_shift) * W) << run_shift) + (j & ((1 << run_shift) - 1));
            if (ch >= n_chunks) continue;                      // wave-uniform
            chunk_max = max(chunk_max, pushed - chunk_mark);   // candidates of the chunk before this one
            chunk_mark = pushed;
            const int r0 = (int)(ch * kRowsPerLoad + kPerLane * lane);
            const unsigned g0 = ((kv[j].x | kTop) - nkr) & kTop, g1 = ((kv[j].y | kTop) - nkr) & kTop;
            const unsigned g2 = ((kv[j].z | kTop) - nkr) & kTop, g3 = ((kv[j].w | kTop) - nkr) & kTop;
            if constexpr (sizeof(KT) == 2) {
                // rows 0..3 keep their flags at bits 15/31/47/63, rows 4..7 move to bits 7/23/39/55
                unsigned long long m = ((unsigned long long)g0 | ((unsigned long long)g1 << 32)) |
                                       (((unsigned long long)g2 | ((unsigned long long)g3 << 32)) >> 8);
                for (;;) {
                    const bool has = m != 0;
                    const int pbit = __ffsll((long long)m) - 1;         // meaningless when !has
                    const int q = (pbit >> 4) + ((pbit & 8) ? 0 : 4);   // row within the lane's eight
                    const unsigned w = q < 4 ? (q < 2 ? kv[j].x : kv[j].y) : (q < 6 ? kv[j].z : kv[j].w);
                    const unsigned kq = (w >> ((q & 1) * 16)) & 0xFFFFu;
                    const int entry = (r0 + q) | (kq == now_key ? (int)0x80000000 : 0);
                    if (!push(has, entry)) break;
                    m &= m - 1;
                }
            } else {
                // byte b of word w is row 4w + b; its flag moves to bit 8b + w
                unsigned m = (g0 >> 7) | (g1 >> 6) | (g2 >> 5) | (g3 >> 4);
                for (;;) {
                    const bool has = m != 0;
                    const int pbit = __ffs((int)m) - 1;
                    const int w = pbit & 7, b = pbit >> 3;
                    const unsigned word = w < 2 ? (w == 0 ? kv[j].x : kv[j].y) : (w == 2 ? kv[j].z : kv[j].w);
                    const unsigned kq = (word >> (8 * b)) & 0xFFu;
                    const int entry = (r0 + 4 * w + b) | (kq == now_key ? (int)0x80000000 : 0);
                    if (!push(has, entry)) break;
                    m &= m - 1;
                }
            }
        }
    }
    if (gw == (n_chunks >> run_shift) % W) { // the wave whose turn the next chunk would be: the table's last, partial chunk
        for (long long r0 = n_chunks * kRowsPerLoad; r0 < n; r0 += kWave) {
            const long long r = r0 + lane;
            const unsigned kq = r < n ? key[r] : 0u;
            push(r < n && kq >= now_key, (int)r | (kq == now_key ? (int)0x80000000 : 0));
        }
    }
    if (lfill > 0) drain(lfill);
    if constexpr (PIPE) { // run the last batches through the remaining steps
        step_c();
        step_b();
        step_c();
    }
    if (st.fill > 0) stage_flush(st, st.fill, out, out_rank, &blk_cursor, lane);
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    if (lane == 0 && namb) atomicAdd(&blk_amb, namb);
    if (lane == 0 && ncand) atomicAdd(&blk_cand, ncand);
    chunk_max = max(chunk_max, pushed - chunk_mark);
    if (lane == 0 && chunk_max) atomicMax(&blk_chunk_max, chunk_max);
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_count[bid] = blk_cursor;
        add_row_stats(summary, blk_live, blk_amb, bid, blk_cand, blk_chunk_max);
    }
    if constexpr (AGG) { // one histogram atomic per (block, hot user); K3 needs the base it returned
        if ((int)threadIdx.x < hot.n) {
            const int cnt = blk_hot_cnt[threadIdx.x];
            blk_hot_base[(long long)bid * kHotMax + threadIdx.x] = cnt ? atomicAdd(&counts[hist_index(hot.user[threadIdx.x], n_users)], cnt) : 0;
        }
    }
}

template <int UNROLL, bool AGG, bool NT, class KT, bool PIPE>
__global__ __launch_bounds__(kK1Threads) void k_scan_keyed(
    const PayRec* __restrict__ pay, const long long* __restrict__ end, const KT* __restrict__ key, long long n,
    long long rows_per_block, long long now, unsigned now_key, long long cutoff, unsigned long long mask, int n_users, int* __restrict__ counts,
    SelRec* __restrict__ sel, int* __restrict__ sel_rank, int* __restrict__ blk_count, Summary* __restrict__ summary,
    DirectSlots direct, HotSet hot, int* __restrict__ blk_hot_base, int run_shift)
{
    scan_keyed_body<UNROLL, AGG, NT, KT, PIPE>(pay, end, key, n, rows_per_block, now, now_key, cutoff, mask, n_users, counts, sel, sel_rank,
                                               blk_count, summary, direct, hot, blk_hot_base, (int)blockIdx.x, (int)gridDim.x, run_shift);
}

// ------------------------------------------------------------------------------------------------ partitioned fast path
//
// For a sparse query (few selected rows per user range) the whole tail collapses into ONE kernel with no host round
// trip: K1P appends every selected record to the partition of its user range (partition = user >> shift, one returning
// atomic on the partition's cursor — the same atomic count as the histogram of the general path), and the tail kernel
// gives each partition to ONE WAVE that sorts its records by (user, start, idx) in registers, derives the per-user
// counts / offsets from the sorted run and the partition totals, and writes counts, offsets and idx directly.
// The host chooses this path from the previous scan's M (records per partition must fit kPartCap with a wide margin);
// a partition that overflows sets a flag and the host reruns the scan on the general path.
__device__ __forceinline__ long long wave_incl_scan(long long v, int lane); // defined with the K2 helpers below

constexpr int kPartCap = 256;   // records per partition the tail wave can sort (4 per lane: keeps the tail under 128 VGPRs)
constexpr int kPartMax = 4096;  // partitions (cursor array size)
constexpr int kPartRange = 256;  // users per partition the tail wave can histogram in its LDS slice

template <int UNROLL, bool NT>
__global__ __launch_bounds__(kK1Threads) void k_scan_live_first_part(
    const long long* __restrict__ start, const long long* __restrict__ end, const int* __restrict__ user,
    const int* __restrict__ disc, long long n, long long rows_per_block, long long now, long long cutoff,
    unsigned long long mask, int n_users, int shift, int* __restrict__ part_cursor, SelRec* __restrict__ part_rec,
    Summary* __restrict__ summary)
{
    __shared__ int live_ring[kK1Waves][kLiveRing];
    __shared__ int blk_live;
    constexpr int kTile = kUnitRows * UNROLL;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) blk_live = 0;
    __syncthreads();
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    long long c1 = c0 + rows_per_block;
    if (c1 > n) c1 = n;
    int* lring = live_ring[wave];
    int lhead = 0, lfill = 0, nlive = 0; // wave-uniform

    auto drain = [&](int cnt) {
        if (lane < cnt) {
            const int row = lring[(lhead + lane) & (kLiveRing - 1)];
            const long long sv = start[row];
            const int dv = disc[row];
            if ((sv >= cutoff) & ((unsigned)dv < 64u) & (((mask >> (dv & 63)) & 1ull) != 0)) {
                const int uv = user[row];
                if ((unsigned)uv < (unsigned)n_users) {
                    const int pt = uv >> shift;
                    const int pos = atomicAdd(&part_cursor[pt], 1);
                    if (pos < kPartCap) {
                        SelRec r;
                        r.start = sv;
                        r.idx = row;
                        r.user = uv;
                        part_rec[(long long)pt * kPartCap + pos] = r;
                    }
                } else {
                    atomicAdd(&summary->bad_rows, 1u);
                }
            }
        }
        lhead = (lhead + cnt) & (kLiveRing - 1);
        lfill -= cnt;
    };
    auto push_live = [&](bool live, int row) {
        const unsigned long long b = __ballot(live);
        if (b == 0) return;
        if (live) lring[(lhead + lfill + prefix_in_ballot(b)) & (kLiveRing - 1)] = row;
        const int c = __popcll(b);
        lfill += c;
        nlive += c;
        __builtin_amdgcn_wave_barrier();
        if (lfill >= kWave) drain(kWave);
        __builtin_amdgcn_wave_barrier();
    };
    for (long long t = c0 + (long long)wave * kTile; t < c1; t += (long long)kTile * kK1Waves) {
        if (t + kTile <= c1) {
            ll2_t e[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j)
                e[j] = stream_load<NT>(reinterpret_cast<const ll2_t*>(end + t + j * kUnitRows + 2 * lane));
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const int r = (int)(t + j * kUnitRows + 2 * lane);
                push_live(e[j].x > now, r);
                push_live(e[j].y > now, r + 1);
            }
        } else {
            const long long t1 = (t + kTile < c1) ? t + kTile : c1;
            for (long long r0 = t; r0 < t1; r0 += kWave) {
                const long long r = r0 + lane;
                push_live(r < t1 && end[r] > now, (int)r);
            }
        }
    }
    if (lfill > 0) drain(lfill);
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    __syncthreads();
    if (threadIdx.x == 0) add_row_stats(summary, blk_live, 0);
}

// bucket of n <= NS records at LDS slots [o, o+n) -> out[0..n) in (start, idx) order: bitonic network in registers,
// every index a compile-time constant (the LDS twin of sort_bucket_regs)
template <int NS>
__device__ __forceinline__ void sort_lds_bucket(const long long* __restrict__ ls, const int* __restrict__ li, int o, int n,
                                                int* __restrict__ out)
{
    long long ks[NS];
    int ki[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const bool in = k < n;
        ks[k] = in ? ls[o + k] : INT64_MAX;
        ki[k] = in ? li[o + k] : INT32_MAX;
    }
#pragma unroll
    for (int k = 2; k <= NS; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool lt = key_less(ks[l], ki[l], ks[i], ki[i]);
                    const bool sw = up ? lt : !lt;
                    const long long s0 = sw ? ks[l] : ks[i], s1 = sw ? ks[i] : ks[l];
                    const int i0 = sw ? ki[l] : ki[i], i1 = sw ? ki[i] : ki[l];
                    ks[i] = s0; ks[l] = s1; ki[i] = i0; ki[l] = i1;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (k < n) out[k] = ki[k];
}

// The tail of the partitioned path: one wave per partition, all in the wave's LDS slice:
//   per-user histogram of the partition's records (LDS atomics; the returned value is the record's slot inside its
//   user's bucket) -> exclusive prefix over the partition's users (counts / offsets go to global memory) -> records
//   placed at their bucket slot in LDS -> every bucket (<= 16 rows) ordered by ONE lane with the register network.
// A sort of the whole partition by (user, start, idx) through the cross-lane network was measured first: 336
// ds_bpermute per wave made the kernel LDS-crossbar bound (25-34 us); bucketing first needs ~200 plain LDS accesses.
// No same-address global atomics anywhere (per-block maxima go to an array; k_publish_summary reduces them).
constexpr int kTailThreads = 512;
constexpr int kTailWaves = kTailThreads / 64;
__global__ __launch_bounds__(kTailThreads) void k_tail_partitions(const int* __restrict__ part_cursor, const SelRec* __restrict__ part_rec,
                                                                  int n_parts, int shift, int n_users, int* __restrict__ counts,
                                                                  long long* __restrict__ offsets, int* __restrict__ out_idx,
                                                                  Summary* __restrict__ summary, unsigned int* __restrict__ blk_max,
                                                                  int4* __restrict__ zero_span, long long zero_vec16)
{
    __shared__ int hist_all[kTailWaves][kPartRange];   // per-user record count, then per-user base inside the partition
    __shared__ long long ls_all[kTailWaves][kPartCap]; // records in bucket order: start ...
    __shared__ int li_all[kTailWaves][kPartCap];       // ... and row index
    __shared__ unsigned int wmax[kTailWaves];
    unsigned int wave_max = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (zero_span) {
        const int4 z = make_int4(0, 0, 0, 0);
        for (long long i = (long long)blockIdx.x * kTailThreads + threadIdx.x; i < zero_vec16; i += (long long)gridDim.x * kTailThreads)
            zero_span[i] = z;
    }
    const int range = 1 << shift;
    int* hist = hist_all[wave];
    long long* ls = ls_all[wave];
    int* li = li_all[wave];
    const int total_waves = gridDim.x * kTailWaves;
    for (int p = blockIdx.x * kTailWaves + wave; p < n_parts; p += total_waves) {
        for (int k = lane; k < range; k += 64) hist[k] = 0;
        // the partition's records are fetched at once, before its count is known (slots past the count are ignored):
        // cursor, cursor prefix and records all travel in the same memory round trip
        const SelRec* rec = part_rec + (long long)p * kPartCap;
        SelRec r4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r4[e] = rec[e * 64 + lane];
        const int raw = part_cursor[p];
        const int cnt = raw < kPartCap ? raw : kPartCap;
        bool overflow = raw > kPartCap;
        // base = records in the partitions before this one: the whole cursor array (<= 4096 ints, zero padded) is
        // fetched with 16 independent 16-B loads per lane, issued together
        long long part = 0;
#pragma unroll 4
        for (int k = 0; k < kPartMax / 256; ++k) {
            const int q0 = (k * 64 + lane) * 4;
            if (k * 256 >= p) break; // wave-uniform: nothing at or past partition p counts
            const int4 v = reinterpret_cast<const int4*>(part_cursor)[k * 64 + lane];
            part += (q0 + 0 < p) ? min(v.x, kPartCap) : 0;
            part += (q0 + 1 < p) ? min(v.y, kPartCap) : 0;
            part += (q0 + 2 < p) ? min(v.z, kPartCap) : 0;
            part += (q0 + 3 < p) ? min(v.w, kPartCap) : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
        const long long base = part;
        const int u0 = p << shift;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 1. slot of every record inside its user's bucket (arrival order; the bucket sort fixes the order)
        int slot[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            slot[e] = 0;
            if (e * 64 + lane < cnt) slot[e] = atomicAdd(&hist[r4[e].user - u0], 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 2. exclusive prefix over the partition's users; counts / offsets out; hist[] becomes the bucket base
        const int per = range >= 64 ? range / 64 : 1;
        const int first = lane * per;
        int mine = 0;
        for (int k = 0; k < per; ++k)
            if (first + k < range) mine += hist[first + k];
        int run = (int)(wave_incl_scan((long long)mine, lane) - mine);
        unsigned int mx = 0;
        for (int k = 0; k < per; ++k) {
            if (first + k < range) {
                const int c = hist[first + k];
                const int u = u0 + first + k;
                if (u < n_users) {
                    counts[u] = c;
                    offsets[u] = base + run;
                }
                hist[first + k] = run | (c << 16); // bucket base (< 256) and size (<= 256) packed for step 4
                run += c;
                mx = max(mx, (unsigned)c);
            }
        }
        wave_max = max(wave_max, mx);
        overflow |= mx > (unsigned)kTinyMax; // a bucket too large for the register network: general path
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. records to their bucket slot
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e * 64 + lane < cnt) {
                const int pos = (hist[r4[e].user - u0] & 0xFFFF) + slot[e];
                ls[pos] = r4[e].start;
                li[pos] = r4[e].idx;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 4. one lane per user: order the bucket, write the rows
        for (int k = 0; k < per; ++k) {
            if (first + k < range) {
                const int h = hist[first + k];
                const int o = h & 0xFFFF, c = h >> 16;
                if (c == 1) out_idx[base + o] = li[o];
                else if (c > 1 && c <= 8) sort_lds_bucket<8>(ls, li, o, c, out_idx + base + o);
                else if (c > 8 && c <= kTinyMax) { // rare (9..16 rows): insertion sort in place, few registers
                    for (int a = 1; a < c; ++a) {
                        const long long sv = ls[o + a];
                        const int iv = li[o + a];
                        int b = a;
                        while (b > 0 && key_less(sv, iv, ls[o + b - 1], li[o + b - 1])) {
                            ls[o + b] = ls[o + b - 1];
                            li[o + b] = li[o + b - 1];
                            --b;
                        }
                        ls[o + b] = sv;
                        li[o + b] = iv;
                    }
                    for (int a = 0; a < c; ++a) out_idx[base + o + a] = li[o + a];
                }
            }
        }
        if (__ballot(overflow) && lane == 0) atomicOr(&summary->pad, 1u); // rare: the host reruns on the general path
        if (p == n_parts - 1 && lane == 0) {
            offsets[n_users] = base + cnt;
            __hip_atomic_store(&summary->m, (unsigned long long)(base + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_max = max(wave_max, (unsigned)__shfl_xor((int)wave_max, o, kWave));
    if (lane == 0) wmax[wave] = wave_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int m = 0;
        for (int w = 0; w < kTailWaves; ++w) m = max(m, wmax[w]);
        blk_max[blockIdx.x] = m;
    }
}

// One block behind the tail: largest bucket = max over the tail blocks' maxima; summary -> mapped host memory.
__global__ __launch_bounds__(256) void k_publish_summary(Summary* __restrict__ summary, const unsigned int* __restrict__ blk_max,
                                                         int n_blk, HostSummary* __restrict__ host, unsigned long long seq)
{
    __shared__ unsigned int wmax[4];
    unsigned int m = 0;
    for (int i = threadIdx.x; i < n_blk; i += 256) m = max(m, blk_max[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, kWave));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    unsigned long long live = 0, amb = 0;
    sum_row_stats(summary, (int)threadIdx.x, live, amb);
    if (threadIdx.x == 0) {
        Summary out = *summary;
        out.live = live;
        out.amb = amb;
        out.cand = 0;
        out.chunk_max = 0;
        out.pad2 = 0;
        out.max_count = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        out.n_seg = out.n_big = out.n_small = 0;
        out.q = 0;
        summary->max_count = out.max_count;
        host->s = out;
        __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------ K2 offsets

__device__ __forceinline__ long long wave_incl_scan(long long v, int lane)
{
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const long long t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide sum of one value per thread (256 threads); result valid in every thread
__device__ __forceinline__ long long block_sum_256(long long v, long long* lds4)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// exclusive prefix over blk[0..nb) by ONE block -> off[nb+1]; total also stored to *total_out (may be null)
__global__ __launch_bounds__(256) void k_block_prefix(const int* __restrict__ blk, int nb, long long* __restrict__ off,
                                                      unsigned long long* __restrict__ total_out)
{
    __shared__ long long wsum[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const long long c = b < nb ? blk[b] : 0;
        const long long incl = wave_incl_scan(c, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        long long wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const long long carry = carry_s;
        if (b < nb) off[b] = carry + wbase + incl - c;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[nb] = carry_s;
        if (total_out) *total_out = (unsigned long long)carry_s;
    }
}

// K2: offsets[u] = exclusive prefix of counts; sort work lists; max bucket; M — in ONE launch.
// Tiles of 256 * UPT users are claimed by ticket (so a block only ever waits for tiles that are already running,
// whatever the dispatch order); a tile publishes its sum as one 8-byte {flag, value} granule with an agent-scope
// store and reads its predecessors' granules with agent-scope loads (no fence needed for a single granule:
// /opt/skills/guides/cdna_hip_programming.md Guideline 16, form R2).  The last block to finish copies the
// summary to mapped host memory (seq last, system scope): the host spins on it instead of paying a D2H copy
// node plus an event wait.  Every block also zeroes its slice of the OTHER slot's histogram span, so the next
// scan needs no memset.
//   UPT    users per thread.  8: few, large tiles (any number of users).  1: one user per thread, used with ORDER.
//   BLOCK  threads per block.  Every block costs two same-address atomics (ticket, done), which serialise at ~11 ns
//          each; the largest bucket travels inside the tile granules instead of through an atomicMax.
//   msg    (ORDER only, optional) the scan's result message [off[0..u_pad] | M | rows[0..cap)] as int32, written here as
//          well, so that a scan whose buckets all fit the direct slots needs no pack kernel before the exchange.
//   ORDER  also do K4's job for buckets of <= kTinyMax rows, from the direct bucket slots: the thread that owns user u
//          loads and orders the bucket in registers while the tile sums of its predecessors arrive, and writes
//          out_idx[offsets[u] ...] as soon as the offset is known.  For a sparse query the whole tail of the scan is
//          this one kernel.  (The host picks it when the user table has direct slots and few enough tiles for the
//          all-predecessors look-back.)
constexpr unsigned long long kTileReady = 1ull << 62;

// One word of the result message a scan writes for the exchange step (see k_pack_results for the layout).  The
// consumer is another queue (or another GPU, over xGMI) that starts as soon as the host has seen the scan's summary, so
// the words are written through to memory instead of waiting in this XCD's L2 for the end of the kernel.
__device__ __forceinline__ void msg_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <int NS>
__device__ __forceinline__ void order_bucket_regs(int n, const BktRec* __restrict__ src, int (&res)[NS])
{
    long long ks[NS];
    int ki[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        BktRec r;
        r.start = INT64_MAX;
        r.idx = INT32_MAX;
        if (k < n) r = src[k];
        ks[k] = r.start;
        ki[k] = r.idx;
    }
#pragma unroll
    for (int k = 2; k <= NS; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool lt = key_less(ks[l], ki[l], ks[i], ki[i]);
                    const bool sw = up ? lt : !lt;
                    const long long s0 = sw ? ks[l] : ks[i], s1 = sw ? ks[i] : ks[l];
                    const int i0 = sw ? ki[l] : ki[i], i1 = sw ? ki[i] : ki[l];
                    ks[i] = s0; ks[l] = s1; ki[i] = i0; ki[l] = i1;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) res[k] = ki[k];
}

// One slot of a device-side list for every lane that calls this (call it from inside the branch that selected the
// lanes): one atomicAdd per wave instead of one per lane — thousands of single appends to one counter serialise.
__device__ __forceinline__ unsigned wave_list_slot(unsigned int* counter, unsigned take = 1u)
{
    const unsigned long long active = __ballot(1);
    const int leader = __ffsll((long long)active) - 1;
    // per-lane `take` (tiles of a big bucket): inclusive scan over the active lanes by a loop over them (rare path)
    unsigned before = 0, total = 0;
    if (take == 1u && __ballot(take != 1u) == 0) {
        before = (unsigned)prefix_in_ballot(active);
        total = (unsigned)__popcll(active);
    } else {
        for (unsigned long long m = active; m; m &= m - 1) {
            const int l = __ffsll((long long)m) - 1;
            const unsigned t = (unsigned)__shfl((int)take, l, kWave);
            if (l < lane_id()) before += t;
            total += t;
        }
    }
    unsigned base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, total);
    base = (unsigned)__shfl((int)base, leader, kWave);
    return base + before;
}

template <int UPT, bool ORDER, int BLOCK>
__device__ __forceinline__ void offsets_body(const int* __restrict__ counts, int* __restrict__ counts_ord, int n_users,
                                                 unsigned long long* __restrict__ tile_pub, ScanCtl* __restrict__ ctl,
                                                 long long* __restrict__ offsets,
                                                 Segment* __restrict__ seg_list, Segment* __restrict__ small_list,
                                                 int* __restrict__ big_list,
                                                 Summary* __restrict__ summary, HostSummary* __restrict__ host,
                                                 unsigned long long seq, int4* __restrict__ zero_span, long long zero_vec16,
                                                 DirectSlots direct, BktRec* __restrict__ bkt,
                                                 int* __restrict__ out_idx, int* __restrict__ msg, int u_pad, long long msg_cap,
                                                 int* __restrict__ msg_counts, const HotSet& hot,
                                                 int hot_thr, int* __restrict__ hot_list, int* __restrict__ over_list, int bid, int nblk)
{
    static_assert(UPT == 8 || UPT == 1, "tile shapes: 2048 users (8 per thread) or one user per thread");
    (void)bkt; // the direct part of outgrown buckets is moved by k_copy_direct, not here
    static_assert(!ORDER || UPT == 1, "the fused order step owns one user per thread");
    constexpr int kTileUsers = BLOCK * UPT;
    constexpr int kWaves = BLOCK / kWave;
    __shared__ long long ldsw[kWaves];
    __shared__ long long wsum[kWaves];
    __shared__ unsigned int wmax[kWaves];
    __shared__ unsigned int wpred[kWaves];
    __shared__ unsigned int tile_s;
    __shared__ long long total_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (zero_span) {
        const int4 z = make_int4(0, 0, 0, 0);
        for (long long i = (long long)bid * BLOCK + threadIdx.x; i < zero_vec16; i += (long long)nblk * BLOCK) zero_span[i] = z;
    }
    if (threadIdx.x == 0) tile_s = atomicAdd(&ctl->ticket, 1u);
    __syncthreads();
    const int tile = (int)tile_s;

    const int u0 = tile * kTileUsers + threadIdx.x * UPT;
    int c[UPT];
    long long tsum = 0;
    // the histogram is read through hist_index (transposed layout); counts_ord gets the same numbers in user order
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        c[k] = (u0 + k < n_users) ? counts[hist_index(u0 + k, n_users)] : 0;
        if (u0 + k < n_users) {
            counts_ord[u0 + k] = c[k];
            if (msg_counts) msg_store(msg_counts + u0 + k, c[k]); // the caller's copy (e.g. mapped host memory), see pie_scan_begin_packed2
        }
    }
    unsigned int local_max = 0;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        tsum += c[k];
        local_max = max(local_max, (unsigned)c[k]);
    }
    const long long incl = wave_incl_scan(tsum, lane);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_max = max(local_max, (unsigned)__shfl_xor((int)local_max, o, kWave));
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wmax[wave] = local_max;
    __syncthreads();
    long long tile_total = 0;
    unsigned int tile_max = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        tile_total += wsum[w];
        tile_max = max(tile_max, wmax[w]);
    }
    // the granule carries the tile's sum (bits 0..30) and its largest bucket (bits 31..61): both are < 2^31
    if (threadIdx.x == 0)
        __hip_atomic_store(&tile_pub[tile], kTileReady | ((unsigned long long)tile_max << 31) | (unsigned long long)tile_total,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // users of this scan's hot set (block-level histogram in K1): all their rows took the staged route, whatever the
    // bucket size turns out to be, so they are never read from the direct slots and always listed for K3 + K4
    bool is_hot[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) is_hot[k] = hot.n > 0 && hot_slot_of(hot, u0 + k) >= 0;

    // ORDER: this thread's bucket, ordered in registers while the predecessors' sums arrive.  Buckets of 2..8 rows go
    // through one 8-slot network (all lanes together); the rare 9..16-row bucket would drag its whole wave through the
    // 16-slot network, so those are ranked cooperatively after the offsets are known (below).
    int res[8];
    if constexpr (ORDER) {
        const int n = is_hot[0] ? 0 : c[0]; // a hot user's rows were staged, not stored in its direct slots
        const BktRec* src = slots_of(direct, u0);
        if (n == 1) res[0] = src[0].idx;
        else if (n >= 2 && n <= 8) order_bucket_regs<8>(n, src, res);
    }

    // base = sum of the tiles in front of this one
    long long part = 0;
    unsigned int pred_max = 0;
    for (int t = threadIdx.x; t < tile; t += BLOCK) {
        unsigned long long v;
        do {
            v = __hip_atomic_load(&tile_pub[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(v & kTileReady)) __builtin_amdgcn_s_sleep(1);
        } while (!(v & kTileReady));
        part += (long long)(v & 0x7FFFFFFFull);
        pred_max = max(pred_max, (unsigned)((v >> 31) & 0x7FFFFFFFull));
    }
    // block-wide sum of `part`, block-wide max of `pred_max`
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_xor(part, o, kWave);
        pred_max = max(pred_max, (unsigned)__shfl_xor((int)pred_max, o, kWave));
    }
    if (lane == 0) { ldsw[wave] = part; wpred[wave] = pred_max; }
    __syncthreads();
    long long base = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        base += ldsw[w];
        pred_max = max(pred_max, wpred[w]);
    }

    long long run = base + incl - tsum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int u = u0 + k;
        if (u < n_users) {
            offsets[u] = run;
            const int n = c[k];
            if constexpr (ORDER) {
                if (msg) msg_store(msg + u, (int)run);
                if (n >= 1 && n <= 8 && !is_hot[k]) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < n) {
                            out_idx[run + i] = res[i];
                            if (msg && run + i < msg_cap) msg_store(msg + u_pad + 2 + run + i, res[i]);
                        }
                }
            }
            // buckets K2 does not order itself: more than kTinyMax rows (without ORDER: every bucket that is not read from
            // the direct slots by k_sort_tiny), and every bucket of a hot user
            const bool listed = n > kTinyMax || (is_hot[k] && n > 0);
            if (listed) {
                // whole bucket still in its direct slots?  then the wave that orders it reads it from there (flags bit 1,
                // the user id above it) and nothing of it was staged; otherwise its direct part joins the staged rest in bkt
                const int dcap = 1 << direct.shift;
                const bool in_direct = direct.p && !is_hot[k] && n <= dcap;
                if (!in_direct) {
                    // listed for k_copy_direct (a wave per bucket moves the direct part behind offsets[u] in bkt); the
                    // count also tells the host that staged records exist, i.e. that K3 has work
                    const unsigned os = wave_list_slot(&summary->n_over);
                    over_list[os] = (direct.p && !is_hot[k]) ? u : -1;
                }
                if (n <= kSmallMax) {
                    const unsigned slot = wave_list_slot(&summary->n_small);
                    Segment sg;
                    sg.pos = run;
                    sg.len = n;
                    sg.flags = in_direct ? ((u << 2) | 2) : 0;
                    small_list[slot] = sg;
                } else if (n <= kSegMax) {
                    const unsigned slot = wave_list_slot(&summary->n_seg);
                    Segment sg;
                    sg.pos = run;
                    sg.len = n;
                    sg.flags = 0;
                    seg_list[slot] = sg;
                } else {
                    const int tiles = (n + kSegMax - 1) / kSegMax;
                    const unsigned slot = wave_list_slot(&summary->n_seg, (unsigned)tiles);
                    for (int t = 0; t < tiles; ++t) {
                        Segment sg;
                        sg.pos = run + (long long)t * kSegMax;
                        sg.len = min(kSegMax, n - t * kSegMax);
                        sg.flags = 1;
                        seg_list[slot + t] = sg;
                    }
                    big_list[wave_list_slot(&summary->n_big)] = u;
                }
            }
            if (hot_thr > 0 && n > hot_thr) { // candidate for the next scan's hot set (see HotSet)
                const unsigned hs = wave_list_slot(&summary->n_hot);
                if (hs < (unsigned)kHotMax) hot_list[hs] = u;
            }
            run += n;
        }
    }
    if constexpr (ORDER) {
        // buckets of 9..16 rows, one at a time with the whole wave: lane i < n holds record i and counts the records that
        // sort before it; that count is its place.  (`run` was advanced past this thread's bucket above: UPT == 1.)
        const int n_mine = c[0];
        const bool mid = u0 < n_users && n_mine > 8 && n_mine <= kTinyMax && !is_hot[0];
        unsigned long long todo = __ballot(mid);
        if (__popcll(todo) > 6) {
            // many such buckets in this wave (users of similar weight sit together): one pass of the 16-slot network
            // for all of them beats ranking them one after the other
            if (mid) {
                const long long ob = run - n_mine;
                int r16[kTinyMax];
                order_bucket_regs<kTinyMax>(n_mine, slots_of(direct, u0), r16);
#pragma unroll
                for (int i = 0; i < kTinyMax; ++i)
                    if (i < n_mine) {
                        out_idx[ob + i] = r16[i];
                        if (msg && ob + i < msg_cap) msg_store(msg + u_pad + 2 + ob + i, r16[i]);
                    }
            }
            todo = 0;
        }
        while (todo) {
            const int src_lane = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int nb = __shfl(n_mine, src_lane, kWave);
            const int ub = __shfl(u0, src_lane, kWave);
            const long long ob = __shfl(run, src_lane, kWave) - nb;
            BktRec r;
            r.start = INT64_MAX;
            r.idx = INT32_MAX;
            if (lane < nb) r = slots_of(direct, ub)[lane];
            int before = 0;
            for (int j = 0; j < nb; ++j) {
                const long long sj = __shfl(r.start, j, kWave);
                const int ij = __shfl(r.idx, j, kWave);
                before += key_less(sj, ij, r.start, r.idx) ? 1 : 0;
            }
            if (lane < nb) {
                out_idx[ob + before] = r.idx;
                if (msg && ob + before < msg_cap) msg_store(msg + u_pad + 2 + ob + before, r.idx);
            }
        }
    }
    if (u0 + UPT >= n_users && u0 < n_users) { // thread holding the last user: it sits in the last tile, which has seen every granule
        offsets[n_users] = run;
        __hip_atomic_store(&summary->m, (unsigned long long)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&summary->max_count, max(pred_max, tile_max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        total_s = run;
    }
    if constexpr (ORDER) {
        // message tail, by the last tile: off[u] = M for the padding users n_users .. u_pad, then the M word
        if (msg && tile == nblk - 1) {
            __syncthreads();
            const long long m_all = total_s;
            for (int u = n_users + threadIdx.x; u <= u_pad + 1; u += BLOCK) msg_store(msg + u, (int)m_all);
        }
    }

    // completion: the last block hands the summary to the host.  Every field was written by device-scope atomics
    // (or the agent-scope store above), each writer's operations are complete before its block's `done` increment
    // (vmcnt drain + barrier), and the reader uses agent-scope loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int is_last;
    if (threadIdx.x == 0) is_last = (atomicAdd(&ctl->done, 1u) == (unsigned)nblk - 1u && host) ? 1 : 0;
    __syncthreads();
    if (is_last && threadIdx.x < 64) {
        unsigned long long live = 0, amb = 0, cand = 0;
        unsigned int chunk_max = 0;
        sum_row_stats(summary, (int)threadIdx.x, live, amb, &cand, &chunk_max); // K1 finished before this kernel started
        if (threadIdx.x == 0) {
            Summary out;
            out.cand = cand;
            out.chunk_max = chunk_max;
            out.pad2 = 0;
            out.m = __hip_atomic_load(&summary->m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.n_seg = __hip_atomic_load(&summary->n_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.n_big = __hip_atomic_load(&summary->n_big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.max_count = __hip_atomic_load(&summary->max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.bad_rows = __hip_atomic_load(&summary->bad_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.q = 0;
            out.live = live;
            out.n_small = __hip_atomic_load(&summary->n_small, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.pad = 0;
            out.amb = amb;
            out.n_hot = __hip_atomic_load(&summary->n_hot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out.n_over = __hip_atomic_load(&summary->n_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host->s = out;
            __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int UPT, bool ORDER, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_offsets(const int* __restrict__ counts, int* __restrict__ counts_ord, int n_users,
                                                 unsigned long long* __restrict__ tile_pub, ScanCtl* __restrict__ ctl,
                                                 long long* __restrict__ offsets,
                                                 Segment* __restrict__ seg_list, Segment* __restrict__ small_list,
                                                 int* __restrict__ big_list,
                                                 Summary* __restrict__ summary, HostSummary* __restrict__ host,
                                                 unsigned long long seq, int4* __restrict__ zero_span, long long zero_vec16,
                                                 DirectSlots direct, BktRec* __restrict__ bkt,
                                                 int* __restrict__ out_idx, int* __restrict__ msg, int u_pad, long long msg_cap,
                                                 int* __restrict__ msg_counts, HotSet hot,
                                                 int hot_thr, int* __restrict__ hot_list, int* __restrict__ over_list)
{
    offsets_body<UPT, ORDER, BLOCK>(counts, counts_ord, n_users, tile_pub, ctl, offsets, seg_list, small_list, big_list, summary, host, seq, zero_span, zero_vec16, direct, bkt, out_idx, msg, u_pad, msg_cap, msg_counts, hot, hot_thr, hot_list, over_list, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------ K1(i+1) + K2(i) in one launch
//
// With two scans in flight, K2 of scan i does not have to sit between two table passes: its inputs are complete when
// K1(i) is, and nothing K1(i+1) touches depends on it (each K2 zeroes the histogram span of the scan AFTER the next).
// So the launch that runs the table pass of scan i+1 carries K2(i) in its first n_tail blocks: the latency-bound chain
// (ticket, look-back, summary hand-off) runs beside the stream of the next pass instead of in front of it.
template <class KT>
struct KeyedArgs {
    const PayRec* pay;
    const long long* end;
    const KT* key;
    long long n, rows_per_block, now;
    unsigned now_key;
    long long cutoff;
    unsigned long long mask;
    int n_users;
    int* counts;
    SelRec* sel;
    int* sel_rank;
    int* blk_count;
    Summary* summary;
    DirectSlots direct;
    HotSet hot;
    int* blk_hot_base;
    int run_shift;   // log2 of the consecutive chunks a wave takes before the round robin moves on (0: fully interleaved)
};
struct OffsetsArgs {
    const int* counts;
    int* counts_ord;
    int n_users;
    unsigned long long* tile_pub;
    ScanCtl* ctl;
    long long* offsets;
    Segment* seg_list;
    Segment* small_list;
    int* big_list;
    Summary* summary;
    HostSummary* host;
    unsigned long long seq;
    int4* zero_span;
    long long zero_vec16;
    DirectSlots direct;
    BktRec* bkt;
    int* out_idx;
    int* msg;
    int u_pad;
    long long msg_cap;
    int* msg_counts;
    HotSet hot;
    int hot_thr;
    int* hot_list;
    int* over_list;
    int n_tail; // blocks [0, n_tail) of the launch do K2 (256 users each), the rest the table pass
};

template <int UNROLL, bool NT, class KT, bool AGG>
__global__ __launch_bounds__(kK1Threads) void k_scan_keyed_with_tail(KeyedArgs<KT> a, OffsetsArgs t)
{
    if ((int)blockIdx.x < t.n_tail) {
        offsets_body<1, true, kK1Threads>(t.counts, t.counts_ord, t.n_users, t.tile_pub, t.ctl, t.offsets, t.seg_list, t.small_list,
                                          t.big_list, t.summary, t.host, t.seq, t.zero_span, t.zero_vec16, t.direct, t.bkt, t.out_idx,
                                          t.msg, t.u_pad, t.msg_cap, t.msg_counts, t.hot, t.hot_thr, t.hot_list, t.over_list, (int)blockIdx.x, t.n_tail);
    } else {
        scan_keyed_body<UNROLL, AGG, NT, KT, false>(a.pay, a.end, a.key, a.n, a.rows_per_block, a.now, a.now_key, a.cutoff, a.mask,
                                                      a.n_users, a.counts, a.sel, a.sel_rank, a.blk_count, a.summary, a.direct, a.hot,
                                                      a.blk_hot_base, (int)blockIdx.x - t.n_tail, (int)gridDim.x - t.n_tail, a.run_shift);
    }
}

// The same for the streaming form (the every-byte scan, dense queries): K2 of scan i in the first blocks of scan i + 1's
// streaming pass, so a steady stream of such scans is one launch each.
struct StreamArgs {
    const long long* start;
    const long long* end;
    const int* user;
    const int* disc;
    long long n, rows_per_block, now, cutoff;
    unsigned long long mask;
    int n_users;
    int* counts;
    SelRec* sel;
    int* sel_rank;
    int* blk_count;
    Summary* summary;
    DirectSlots direct;
};

template <bool LATE_U>
__global__ __launch_bounds__(kK1Threads) void k_scan_compact_with_tail(StreamArgs a, OffsetsArgs t)
{
    if ((int)blockIdx.x < t.n_tail) {
        offsets_body<1, true, kK1Threads>(t.counts, t.counts_ord, t.n_users, t.tile_pub, t.ctl, t.offsets, t.seg_list, t.small_list, t.big_list, t.summary,
                                          t.host, t.seq, t.zero_span, t.zero_vec16, t.direct, t.bkt, t.out_idx, t.msg, t.u_pad, t.msg_cap,
                                          t.msg_counts, t.hot, t.hot_thr, t.hot_list, t.over_list, (int)blockIdx.x, t.n_tail);
    } else {
        scan_compact_body<4, true, LATE_U>(a.start, a.end, a.user, a.disc, a.n, a.rows_per_block, a.now, a.cutoff, a.mask, a.n_users, a.counts, a.sel,
                                           a.sel_rank, a.blk_count, a.summary, a.direct, nullptr, (int)blockIdx.x - t.n_tail);
    }
}

// ------------------------------------------------------------------------------------------------ batched scan: Q queries, one table pass
//
// SURVEY.md section 7 ("batch many queries per launch") / the north_star's "batched GPU scan": a server answers many
// feed requests, each with its own `now` (millisecond clock), `cutoff` and discipline mask.  Requests that arrive
// together select almost the same rows, so the batch works on the UNION of their selections:
//   table pass   streams the key column once, takes the rows that are candidates for ANY query (key >= the smallest
//                key(now)), gathers each candidate's payload record once, evaluates the Q predicates into a Q-bit query
//                mask, and — if any bit is set — issues ONE histogram atomic (rank in the user's union bucket) and ONE
//                bucket-slot store {start, row, query mask}: the same atomics and stores as a single query, whatever Q
//   offsets      one thread per user loads and orders its union bucket once, counts the rows of every query in it, runs Q
//                prefix scans side by side (tile granules per query), and writes Q sets of counts / offsets / row lists.
// (A first version kept one histogram and one bucket set per query: every extra query then cost its own ~3 x 10^5
// returning atomics, ~10 us at the chip's scattered-atomic rate, plus its own offsets kernel — profiles/r02_b_*.)
// There is no staged route here: a user whose union bucket outgrows its slots is reported (Summary::n_over) and the host
// reruns the batch's queries on the general path, bit for bit the same result; the slot capacity then grows (up to 64).
constexpr int kBatchMax = 64;
constexpr int kUnionShiftMax = 6; // union bucket slots per user: 16 .. 64 (one wave orders a bucket by ranking)

// The Q predicates of a candidate row in O(log Q): `end > now_q` and `start >= cutoff_q` are monotone in the query's scalar,
// so with the batch's `now` values and cutoffs sorted, the queries a row is live for / in the window of are a PREFIX of the
// sorted order — two binary searches and two prefix-mask lookups — and the discipline conjunct is one lookup in a table
// "which queries accept discipline d".  qmask = live[rank(end)] & win[rank(start)] & disc[d].  (Round 2 looped over the
// queries: 64 of them cost 0.101 ms per pass against 0.064 for 16.)  Built on the host, shipped in the kernel arguments,
// copied to LDS by every block.  Queries that fall back are simply not in the tables.
struct BatchTables {
    long long now[kBatchMax];               // ascending; unused entries INT64_MAX
    long long cutoff[kBatchMax];            // ascending; unused entries INT64_MAX
    unsigned long long live[kBatchMax + 1]; // live[r]: the queries with the r smallest `now`
    unsigned long long win[kBatchMax + 1];  // win[r]: the queries with the r smallest cutoff
    unsigned long long disc[64];            // disc[d]: the queries whose mask has bit d
    unsigned nk[kBatchMax];                 // key(now) in the order of now[] (monotone, so ascending too); unused entries ~0
};
static_assert(sizeof(BatchTables) % 4 == 0, "copied to LDS word by word");

template <class KT>
struct BatchScanArgs {
    const PayRec* pay;
    const long long* end;
    const KT* key;
    long long n, rows_per_block;
    int n_users;
    int n_q;
    unsigned min_key;          // smallest now_key of the batch: a row below it is dead for every query
    int dshift;                // log2 of the union bucket's slot capacity
    int* counts;               // union histogram (transposed user order, hist_index)
    Summary* summary;          // the batch's summary: bad rows and the row statistics of the pass
    BktRec* direct;            // union bucket slots, (1 << dshift) per user; BktRec::pad = the queries (0..31) that selected the row
    unsigned* direct_hi;       // queries 32..63 of every slot (written when n_q > 32)
    int run_shift;             // see KeyedArgs
    BatchTables tab;
};

// entries of an ascending 64-entry table that are < x (LT) or <= x (!LT): branchless, seven probes
template <bool LT, class T>
__device__ __forceinline__ int rank_in_64(const T* tab, T x)
{
    int r = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const T v = tab[r + s - 1];
        r += (LT ? (v < x) : (v <= x)) ? s : 0;
    }
    const T v = tab[r];
    r += (LT ? (v < x) : (v <= x)) ? 1 : 0;
    return r;
}

template <int UNROLL, bool NT, class KT>
__device__ __forceinline__ void scan_batch_body(const BatchScanArgs<KT>& a, int bid, int n_scan_blocks)
{
    __shared__ int ring_row[kK1Waves][kLiveRing];
    __shared__ int ring_key[kK1Waves][kLiveRing];
    __shared__ int blk_cand;
    __shared__ int blk_chunk_max;
    __shared__ BatchTables tab;
    constexpr int kLogUnroll = UNROLL >= 8 ? 3 : UNROLL >= 4 ? 2 : UNROLL >= 2 ? 1 : 0;
    const int run_shift = a.run_shift < kLogUnroll ? (a.run_shift < 0 ? 0 : a.run_shift) : kLogUnroll;
    int pushed = 0, chunk_max = 0, chunk_mark = 0;
    constexpr int kPerLane = 16 / (int)sizeof(KT);
    constexpr int kRowsPerLoad = kPerLane * kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_cand = 0; blk_chunk_max = 0; }
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(&a.tab);
        unsigned* dst = reinterpret_cast<unsigned*>(&tab);
        for (int i = threadIdx.x; i < (int)(sizeof(BatchTables) / 4); i += kK1Threads) dst[i] = src[i];
    }
    __syncthreads();
    int* rrow = ring_row[wave];
    int* rkey = ring_key[wave];
    int lhead = 0, lfill = 0, ncand = 0; // wave-uniform
    const int nq = a.n_q;
    const int cap = 1 << a.dshift;

    // evaluate `cnt` queued candidates (cnt <= 64), one per lane, against every query of the batch
    auto drain = [&](int cnt) {
        const bool valid = lane < cnt;
        int row = 0;
        unsigned key = 0;
        PayRec pr;
        pr.start = 0; pr.user = 0; pr.disc = -1;
        if (valid) {
            const int slot = (lhead + lane) & (kLiveRing - 1);
            row = rrow[slot];
            key = (unsigned)rkey[slot];
            pr = a.pay[row]; // ONE gather per candidate, shared by all queries
        }
        // liveness: the queries whose key(now) lies below the row's key form a prefix of the sorted order; a row whose key
        // EQUALS some query's is ranked by its 8-byte `end` instead (key() is monotone, so that is exact for every query)
        int r = rank_in_64<true>(tab.nk, key);
        const bool amb = valid && r < kBatchMax && tab.nk[r < kBatchMax ? r : 0] == key;
        if (amb) r = rank_in_64<true>(tab.now, a.end[row]);
        const int w = rank_in_64<false>(tab.cutoff, pr.start);
        unsigned long long qmask = 0;
        if (valid && (unsigned)pr.disc < 64u) qmask = tab.live[r] & tab.win[w] & tab.disc[pr.disc & 63];
        if (qmask && (unsigned)pr.user >= (unsigned)a.n_users) {
            atomicAdd(&a.summary->bad_rows, 1u);
            qmask = 0;
        }
        if (qmask) {
            const int rank = atomicAdd(&a.counts[hist_index(pr.user, a.n_users)], 1);
            if (rank < cap) {
                BktRec rec;
                rec.start = pr.start;
                rec.idx = row;
                rec.pad = (int)(unsigned)qmask;
                a.direct[((long long)pr.user << a.dshift) + rank] = rec;
                if (nq > 32) a.direct_hi[((long long)pr.user << a.dshift) + rank] = (unsigned)(qmask >> 32);
            }
        }
        ncand += cnt;
        lhead = (lhead + cnt) & (kLiveRing - 1);
        lfill -= cnt;
    };
    auto push = [&](bool cand, int row, unsigned key) {
        const unsigned long long b = __ballot(cand);
        if (b == 0) return false;
        if (cand) {
            const int slot = (lhead + lfill + prefix_in_ballot(b)) & (kLiveRing - 1);
            rrow[slot] = row;
            rkey[slot] = (int)key;
        }
        lfill += __popcll(b);
        pushed += __popcll(b);
        __builtin_amdgcn_wave_barrier();
        if (lfill >= kWave) drain(kWave);
        __builtin_amdgcn_wave_barrier();
        return true;
    };

    // SWAR candidate test against the batch's smallest key(now) (see scan_keyed_body)
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr unsigned kTop = sizeof(KT) == 2 ? 0x80008000u : 0x80808080u;
    const unsigned mk = a.min_key;
    const unsigned nkr = sizeof(KT) == 2 ? (mk | (mk << 16)) : mk * 0x01010101u;
    // rows dealt to the launch's waves in chunks of one load per lane, round robin (see scan_keyed_body)
    const long long n_chunks = a.n / kRowsPerLoad;
    const long long W = (long long)n_scan_blocks * kK1Waves;
    const long long gw = (long long)bid * kK1Waves + wave;
    for (long long cb = gw << run_shift; cb < n_chunks; cb += W * UNROLL) {
        u4_t kv[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const long long ch = cb + (((long long)(j >> run_shift) * W) << run_shift) + (j & ((1 << run_shift) - 1));
            kv[j] = (u4_t){0u, 0u, 0u, 0u};
            if (ch < n_chunks) kv[j] = stream_load<NT>(reinterpret_cast<const u4_t*>(a.key + ch * kRowsPerLoad + kPerLane * lane));
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const long long ch = cb + (((long long)(j >> run_shift) * W) << run_shift) + (j & ((1 << run_shift) - 1));
            if (ch >= n_chunks) continue;
            chunk_max = max(chunk_max, pushed - chunk_mark);
            chunk_mark = pushed;
            const int r0 = (int)(ch * kRowsPerLoad + kPerLane * lane);
            const unsigned g0 = ((kv[j].x | kTop) - nkr) & kTop, g1 = ((kv[j].y | kTop) - nkr) & kTop;
            const unsigned g2 = ((kv[j].z | kTop) - nkr) & kTop, g3 = ((kv[j].w | kTop) - nkr) & kTop;
            if constexpr (sizeof(KT) == 2) {
                unsigned long long m = ((unsigned long long)g0 | ((unsigned long long)g1 << 32)) |
                                       (((unsigned long long)g2 | ((unsigned long long)g3 << 32)) >> 8);
                for (;;) {
                    const bool has = m != 0;
                    const int pbit = __ffsll((long long)m) - 1;
                    const int q = (pbit >> 4) + ((pbit & 8) ? 0 : 4);
                    const unsigned w = q < 4 ? (q < 2 ? kv[j].x : kv[j].y) : (q < 6 ? kv[j].z : kv[j].w);
                    const unsigned kq = (w >> ((q & 1) * 16)) & 0xFFFFu;
                    if (!push(has, r0 + q, kq)) break;
                    m &= m - 1;
                }
            } else {
                unsigned m = (g0 >> 7) | (g1 >> 6) | (g2 >> 5) | (g3 >> 4);
                for (;;) {
                    const bool has = m != 0;
                    const int pbit = __ffs((int)m) - 1;
                    const int w = pbit & 7, b = pbit >> 3;
                    const unsigned word = w < 2 ? (w == 0 ? kv[j].x : kv[j].y) : (w == 2 ? kv[j].z : kv[j].w);
                    const unsigned kq = (word >> (8 * b)) & 0xFFu;
                    if (!push(has, r0 + 4 * w + b, kq)) break;
                    m &= m - 1;
                }
            }
        }
    }
    if (gw == (n_chunks >> run_shift) % W) {
        for (long long r0 = n_chunks * kRowsPerLoad; r0 < a.n; r0 += kWave) {
            const long long r = r0 + lane;
            const unsigned kq = r < a.n ? a.key[r] : 0u;
            push(r < a.n && kq >= mk, (int)r, kq);
        }
    }
    if (lfill > 0) drain(lfill);
    if (lane == 0 && ncand) atomicAdd(&blk_cand, ncand);
    chunk_max = max(chunk_max, pushed - chunk_mark);
    if (lane == 0 && chunk_max) atomicMax(&blk_chunk_max, chunk_max);
    __syncthreads();
    if (threadIdx.x == 0) add_row_stats(a.summary, 0, 0, bid, blk_cand, blk_chunk_max);
}

// offsets + order kernel of a batch: blocks of 256 users.  The batch's PRIMARY result is the union: per user the rows that any
// query selected, in (start, row) order, with a query mask per row —
//     uoff[U+1] | urows[Mu] | umask_lo[Mu] (| umask_hi[Mu] when the batch holds more than 32 queries)
// Feed(q, u) = the rows of urows[uoff[u] : uoff[u+1]] whose mask has bit q, in that order.  ONE prefix scan (the union counts),
// one ordered copy of every bucket, whatever Q; per-query counts / offsets / row lists are materialised from it only on request
// (k_mat_*), and the multi-GPU exchange message is the same arrays as int32 words, written here when the caller asked for it.
// (Round 2 wrote Q full sets of counts / offsets / row lists in this kernel: 47 MB per 16-query batch at cfg3, 0.19 of peak.)
constexpr int kMqSlots = 64; // per-query selected-row totals: blocks add to slot (tile mod 64), the last block sums the slots

struct BatchHost {           // mapped pinned host memory, one per batch slot: written by the tail's last block, seq last
    Summary s;               // m = Mu (union rows), max_count = largest union bucket
    unsigned long long mq[kBatchMax]; // selected rows per query
    unsigned long long seq;
};

struct UnionTailArgs {
    int n_q;
    int n_users;
    int tiles;                 // blocks (256 users each)
    int dshift;
    char* span;                // the batch's span: union histogram | tile granules | .. | ScanCtl | Summary + row statistics | mq slots
    long long tiles_off, ctl_off, summary_off, mq_off; // byte offsets inside a span
    char* zero_span;           // the span the batch after the next will use: zeroed here
    long long zero_total16;    // 16-byte vectors of a span
    const BktRec* direct;      // union bucket slots (BktRec::pad = query bits 0..31)
    const unsigned* direct_hi; // query bits 32..63 of every slot (batches of more than 32 queries)
    long long* uoff;           // [n_users + 1]
    int* urows;                // [n_users << dshift]
    unsigned* umlo;
    unsigned* umhi;
    BatchHost* host;
    unsigned long long seq;
    int* msg;                  // optional union message (device-visible): [uoff[0..u_pad] | Mu | rows[cap) | mask_lo[cap) | mask_hi[cap) if n_q > 32]
    int u_pad;
    long long msg_cap;
};

__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}

// value of the lane R places to the right inside the caller's row of 16 lanes (DPP row_ror: no LDS, no bpermute)
template <int R>
__device__ __forceinline__ int ror16(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + R, 0xF, 0xF, false);
}
// how many of the (up to 16) records held by the lanes of my row of 16 sort before mine by (start, idx); lanes without a record
// hold (INT64_MAX, INT32_MAX) and never count
template <int R>
__device__ __forceinline__ int rank16_step(long long s, int i)
{
    const int lo = ror16<R>((int)(unsigned)(unsigned long long)s), hi = ror16<R>((int)((unsigned long long)s >> 32));
    const long long so = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    const int io = ror16<R>(i);
    return key_less(so, io, s, i) ? 1 : 0;
}
__device__ __forceinline__ int rank_in_row16(long long s, int i)
{
    return rank16_step<1>(s, i) + rank16_step<2>(s, i) + rank16_step<3>(s, i) + rank16_step<4>(s, i) + rank16_step<5>(s, i) +
           rank16_step<6>(s, i) + rank16_step<7>(s, i) + rank16_step<8>(s, i) + rank16_step<9>(s, i) + rank16_step<10>(s, i) +
           rank16_step<11>(s, i) + rank16_step<12>(s, i) + rank16_step<13>(s, i) + rank16_step<14>(s, i) + rank16_step<15>(s, i);
}

// HI = the batch holds more than 32 queries (a second mask word per row)
template <bool HI>
__device__ __forceinline__ void union_tail_body(const UnionTailArgs& t, int gbid)
{
    constexpr int BLOCK = kK1Threads;
    constexpr int kWaves = kK1Waves;
    __shared__ long long s_sum[kWaves];
    __shared__ unsigned int s_max[kWaves];
    __shared__ long long s_part[kWaves];
    __shared__ unsigned int s_pmax[kWaves];
    __shared__ unsigned int s_mq[kWaves][kBatchMax];
    __shared__ long long s_total;
    __shared__ unsigned int tile_s;
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk = t.tiles, U = t.n_users, nq = t.n_q;
    if (t.zero_span) {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* zs = reinterpret_cast<int4*>(t.zero_span);
        for (long long i = (long long)gbid * BLOCK + threadIdx.x; i < t.zero_total16; i += (long long)nblk * BLOCK) zs[i] = z;
    }
    ScanCtl* ctl = reinterpret_cast<ScanCtl*>(t.span + t.ctl_off);
    Summary* sum = reinterpret_cast<Summary*>(t.span + t.summary_off);
    unsigned int* mq_slots = reinterpret_cast<unsigned int*>(t.span + t.mq_off);
    if (threadIdx.x == 0) tile_s = atomicAdd(&ctl->ticket, 1u);
    __syncthreads();
    const int tile = (int)tile_s;
    const int u = tile * BLOCK + (int)threadIdx.x;
    const bool in_u = u < U;
    const int* counts = reinterpret_cast<const int*>(t.span);
    const int cap = 1 << t.dshift;
    const int n_raw = in_u ? counts[hist_index(u, U)] : 0;
    const int nn = n_raw < cap ? n_raw : cap;
    const long long slot0 = (long long)(in_u ? u : 0) << t.dshift;
    const BktRec* src = t.direct + slot0;

    // the user's union bucket: up to 8 rows ordered in this thread's registers; 9 .. 64 rows by the whole wave (below)
    long long ks[8];
    int ki[8];
    unsigned km[8], kh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        BktRec r;
        r.start = INT64_MAX;
        r.idx = INT32_MAX;
        r.pad = 0;
        unsigned h = 0;
        if (nn <= 8 && k < nn) {
            r = src[k];
            if constexpr (HI) h = t.direct_hi[slot0 + k];
        }
        ks[k] = r.start;
        ki[k] = r.idx;
        km[k] = (unsigned)r.pad;
        kh[k] = h;
    }
    if (nn >= 2 && nn <= 8) {
#pragma unroll
        for (int k = 2; k <= 8; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const bool up = (i & k) == 0;
                        const bool lt = key_less(ks[l], ki[l], ks[i], ki[i]);
                        const bool sw = up ? lt : !lt;
                        const long long s0 = sw ? ks[l] : ks[i], s1 = sw ? ks[i] : ks[l];
                        const int i0 = sw ? ki[l] : ki[i], i1 = sw ? ki[i] : ki[l];
                        const unsigned m0 = sw ? km[l] : km[i], m1 = sw ? km[i] : km[l];
                        ks[i] = s0; ks[l] = s1; ki[i] = i0; ki[l] = i1; km[i] = m0; km[l] = m1;
                        if constexpr (HI) {
                            const unsigned h0 = sw ? kh[l] : kh[i], h1 = sw ? kh[i] : kh[l];
                            kh[i] = h0; kh[l] = h1;
                        }
                    }
                }
            }
        }
    }
    // ONE prefix scan: the union counts
    const int incl = wave_incl_scan_i32(nn, lane);
    {
        unsigned mx = (unsigned)nn;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, kWave));
        if (lane == 63) s_sum[wave] = incl;
        if (lane == 0) s_max[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) { // the tile's granule: its sum (bits 0..30) and largest bucket (bits 31..61)
        long long tot = 0;
        unsigned mx = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { tot += s_sum[w]; mx = max(mx, s_max[w]); }
        unsigned long long* pub = reinterpret_cast<unsigned long long*>(t.span + t.tiles_off);
        __hip_atomic_store(&pub[tile], kTileReady | ((unsigned long long)mx << 31) | (unsigned long long)tot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // while the predecessors' granules arrive: selected rows per query (what pie_scan_batch_finish reports).  Per lane the
    // (<= 8) masks of its bucket are added up bit-sliced — planes b0..b3: bit q of plane i = bit i of "how many of my rows
    // query q selected" — by a carry-save adder tree, no loop over queries; then lane q of the wave collects query q's total
    // from four ballots per query.  No branch inside the loop: the chains of consecutive queries interleave (a first version
    // with one ballot per bucket slot and query, each behind a wave-uniform branch, cost 20 of the kernel's 50 us at Q = 64).
    const bool mid = nn > 8;
    unsigned acc = 0;
    {
        auto fa = [](unsigned a, unsigned b, unsigned c, unsigned& carry) { const unsigned x = a ^ b; carry = (a & b) | (c & x); return x ^ c; };
        unsigned pl[4], ph[4];
        {
            unsigned c1, c2, c3, c4, c5, c6;
            const unsigned s1 = fa(km[0], km[1], km[2], c1), s2 = fa(km[3], km[4], km[5], c2);
            const unsigned s3 = km[6] ^ km[7];
            c3 = km[6] & km[7];
            pl[0] = fa(s1, s2, s3, c4);
            const unsigned tw = fa(c1, c2, c3, c5);
            pl[1] = tw ^ c4;
            c6 = tw & c4;
            pl[2] = c5 ^ c6;
            pl[3] = c5 & c6;
        }
        if constexpr (HI) {
            unsigned c1, c2, c3, c4, c5, c6;
            const unsigned s1 = fa(kh[0], kh[1], kh[2], c1), s2 = fa(kh[3], kh[4], kh[5], c2);
            const unsigned s3 = kh[6] ^ kh[7];
            c3 = kh[6] & kh[7];
            ph[0] = fa(s1, s2, s3, c4);
            const unsigned tw = fa(c1, c2, c3, c5);
            ph[1] = tw ^ c4;
            c6 = tw & c4;
            ph[2] = c5 ^ c6;
            ph[3] = c5 & c6;
        } else {
            ph[0] = ph[1] = ph[2] = ph[3] = 0;
        }
        const int nq_lo = nq < 32 ? nq : 32;
#pragma unroll 4
        for (int q = 0; q < nq_lo; ++q) {
            const unsigned c = (unsigned)__popcll(__ballot((pl[0] >> q) & 1u)) + 2u * (unsigned)__popcll(__ballot((pl[1] >> q) & 1u)) +
                               4u * (unsigned)__popcll(__ballot((pl[2] >> q) & 1u)) + 8u * (unsigned)__popcll(__ballot((pl[3] >> q) & 1u));
            if (lane == q) acc = c;
        }
        if constexpr (HI) {
#pragma unroll 4
            for (int q = 32; q < nq; ++q) {
                const unsigned c = (unsigned)__popcll(__ballot((ph[0] >> (q - 32)) & 1u)) + 2u * (unsigned)__popcll(__ballot((ph[1] >> (q - 32)) & 1u)) +
                                   4u * (unsigned)__popcll(__ballot((ph[2] >> (q - 32)) & 1u)) + 8u * (unsigned)__popcll(__ballot((ph[3] >> (q - 32)) & 1u));
                if (lane == q) acc = c;
            }
        }
    }
    // base = sum of the granules of the tiles in front of this one
    long long part = 0;
    unsigned pmax = 0;
    {
        const unsigned long long* pub = reinterpret_cast<const unsigned long long*>(t.span + t.tiles_off);
        for (int tt = threadIdx.x; tt < tile; tt += BLOCK) {
            unsigned long long v;
            do {
                v = __hip_atomic_load(&pub[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(v & kTileReady)) __builtin_amdgcn_s_sleep(1);
            } while (!(v & kTileReady));
            part += (long long)(v & 0x7FFFFFFFull);
            pmax = max(pmax, (unsigned)((v >> 31) & 0x7FFFFFFFull));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_xor(part, o, kWave);
        pmax = max(pmax, (unsigned)__shfl_xor((int)pmax, o, kWave));
    }
    if (lane == 0) { s_part[wave] = part; s_pmax[wave] = pmax; }
    __syncthreads();
    long long run = incl - nn;
    {
        long long base = 0;
        unsigned pm = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { base += s_part[w]; pm = max(pm, s_pmax[w]); }
        pmax = pm;
        run += base;
        for (int w = 0; w < wave; ++w) run += s_sum[w];
    }
    int* msg_rows = t.msg ? t.msg + t.u_pad + 2 : nullptr;
    int* msg_lo = t.msg ? msg_rows + t.msg_cap : nullptr;
    int* msg_hi = t.msg ? msg_lo + t.msg_cap : nullptr;
    if (in_u) {
        t.uoff[u] = run;
        if (t.msg) msg_store(t.msg + u, (int)run);
        if (nn >= 1 && nn <= 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nn) {
                    const long long pos = run + k;
                    t.urows[pos] = ki[k];
                    t.umlo[pos] = km[k];
                    if constexpr (HI) t.umhi[pos] = kh[k];
                    if (t.msg && pos < t.msg_cap) {
                        msg_store(msg_rows + pos, ki[k]);
                        msg_store(msg_lo + pos, (int)km[k]);
                        if constexpr (HI) msg_store(msg_hi + pos, (int)kh[k]);
                    }
                }
        }
    }
    // buckets of 9 .. 16 rows, FOUR at a time: each takes a row of 16 lanes, lane l of the row holds record l and finds its place
    // by comparing against the other fifteen through DPP row rotations (a heterogeneous batch — several role masks — doubles the
    // union: a fifth of the users land here, and one at a time by the whole wave they cost more than the rest of the kernel)
    {
        unsigned long long todo = __ballot(mid && in_u && nn <= 16);
        const int row = lane >> 4, l = lane & 15;
        while (todo) {
            // the row-th pending bucket of this round (rows without one idle)
            unsigned long long t2 = todo;
            int mine = -1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int sl = t2 ? __ffsll((long long)t2) - 1 : -1;
                if (g == row) mine = sl;
                t2 &= t2 - 1;      // (0 & anything stays 0)
            }
            todo = t2;
            const int srcl = mine < 0 ? 0 : mine;
            int nb = __shfl(nn, srcl, kWave); // (every lane takes part in every shuffle: a source lane must be active)
            if (mine < 0) nb = 0;
            const int ub = __shfl(u, srcl, kWave);
            const long long rq = __shfl(run, srcl, kWave);
            BktRec r;
            r.start = INT64_MAX;
            r.idx = INT32_MAX;
            r.pad = 0;
            unsigned hi = 0;
            if (l < nb) {
                r = (t.direct + ((long long)ub << t.dshift))[l];
                if constexpr (HI) hi = t.direct_hi[((long long)ub << t.dshift) + l];
            }
            const int rank = rank_in_row16(r.start, r.idx);
            if (l < nb) {
                const long long pos = rq + rank;
                t.urows[pos] = r.idx;
                t.umlo[pos] = (unsigned)r.pad;
                if constexpr (HI) t.umhi[pos] = hi;
                if (t.msg && pos < t.msg_cap) {
                    msg_store(msg_rows + pos, r.idx);
                    msg_store(msg_lo + pos, r.pad);
                    if constexpr (HI) msg_store(msg_hi + pos, (int)hi);
                }
            }
            // per-query totals of these (up to 64) records: lane q collects query q's
#pragma unroll 4
            for (int q = 0; q < (nq < 32 ? nq : 32); ++q) {
                const unsigned c = (unsigned)__popcll(__ballot(((unsigned)r.pad >> q) & 1u));
                if (lane == q) acc += c;
            }
            if constexpr (HI) {
#pragma unroll 4
                for (int q = 32; q < nq; ++q) {
                    const unsigned c = (unsigned)__popcll(__ballot((hi >> (q - 32)) & 1u));
                    if (lane == q) acc += c;
                }
            }
        }
    }
    // still larger buckets (17 .. 64 rows: slot capacities 32 / 64): lane i < n holds record i and counts the records that sort before it
    {
        unsigned long long todo = __ballot(mid && in_u && nn > 16);
        while (todo) {
            const int src_lane = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int nb = __shfl(nn, src_lane, kWave);
            const int ub = __shfl(u, src_lane, kWave);
            const long long rq = __shfl(run, src_lane, kWave);
            BktRec r;
            r.start = INT64_MAX;
            r.idx = INT32_MAX;
            r.pad = 0;
            unsigned hi = 0;
            if (lane < nb) {
                r = (t.direct + ((long long)ub << t.dshift))[lane];
                if constexpr (HI) hi = t.direct_hi[((long long)ub << t.dshift) + lane];
            }
            int rank = 0;
            for (int j = 0; j < nb; ++j) {
                const long long sj = __shfl(r.start, j, kWave);
                const int ij = __shfl(r.idx, j, kWave);
                rank += key_less(sj, ij, r.start, r.idx) ? 1 : 0;
                // lane q counts the rows of this bucket that query q selected (the per-query totals)
                unsigned mj = (unsigned)__shfl(r.pad, j, kWave);
                if constexpr (HI) {
                    const unsigned hj = (unsigned)__shfl((int)hi, j, kWave);
                    mj = lane >= 32 ? hj : mj;
                }
                acc += (mj >> (lane & 31)) & 1u;
            }
            if (lane < nb) {
                const long long pos = rq + rank;
                t.urows[pos] = r.idx;
                t.umlo[pos] = (unsigned)r.pad;
                if constexpr (HI) t.umhi[pos] = hi;
                if (t.msg && pos < t.msg_cap) {
                    msg_store(msg_rows + pos, r.idx);
                    msg_store(msg_lo + pos, r.pad);
                    if constexpr (HI) msg_store(msg_hi + pos, (int)hi);
                }
            }
        }
    }
    // the block's per-query totals -> the slot of this tile
    s_mq[wave][lane] = acc;
    __syncthreads();
    if ((int)threadIdx.x < nq) {
        unsigned tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += s_mq[w][threadIdx.x];
        if (tot) atomicAdd(&mq_slots[(tile & (kMqSlots - 1)) * kBatchMax + (int)threadIdx.x], tot);
    }
    if (in_u && n_raw > cap) (void)wave_list_slot(&sum->n_over); // a bucket outgrew its slots: the host reruns the queries
    if (in_u && u == U - 1) { // the thread holding the last user sits in the last tile, which has seen every granule
        unsigned tmax = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tmax = max(tmax, s_max[w]);
        const long long m_all = run + nn;
        t.uoff[U] = m_all;
        __hip_atomic_store(&sum->m, (unsigned long long)m_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sum->max_count, max(pmax, tmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_total = m_all;
    }
    if (t.msg && tile == nblk - 1) { // message tail: uoff[u] = Mu for the padding users, then the Mu word
        __syncthreads();
        const long long m_all = s_total;
        for (int uu = U + threadIdx.x; uu <= t.u_pad + 1; uu += BLOCK) msg_store(t.msg + uu, (int)m_all);
    }
    // completion: the last block hands the summary and the per-query totals to the host (see offsets_body)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) is_last = (atomicAdd(&ctl->done, 1u) == (unsigned)nblk - 1u && t.host) ? 1 : 0;
    __syncthreads();
    if (is_last) {
        // per-query totals: 64 slots x n_q counters.  All 256 threads read (16 slots each, INDEPENDENT loads: summed straight
        // into one register the compiler waits for each of them — 64 dependent L2 round trips were 14 of the kernel's 38 us)
        {
            const int q = (int)threadIdx.x & 63, g = (int)threadIdx.x >> 6;
            unsigned v[kMqSlots / kWaves];
#pragma unroll
            for (int i = 0; i < kMqSlots / kWaves; ++i)
                v[i] = q < nq ? __hip_atomic_load(&mq_slots[(g * (kMqSlots / kWaves) + i) * kBatchMax + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            unsigned tot = 0;
#pragma unroll
            for (int i = 0; i < kMqSlots / kWaves; ++i) tot += v[i];
            s_mq[g][q] = tot;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            unsigned long long live = 0, amb = 0, cand = 0;
            unsigned int chunk_max = 0;
            sum_row_stats(sum, (int)threadIdx.x, live, amb, &cand, &chunk_max);
            unsigned long long mq = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) mq += s_mq[w][threadIdx.x];
            t.host->mq[threadIdx.x] = mq;
            if (threadIdx.x == 0) {
                Summary out;
                out.m = __hip_atomic_load(&sum->m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.n_seg = out.n_big = out.n_small = 0;
                out.max_count = __hip_atomic_load(&sum->max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.bad_rows = __hip_atomic_load(&sum->bad_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.q = 0;
                out.live = 0;
                out.pad = 0;
                out.amb = 0;
                out.n_hot = 0;
                out.n_over = __hip_atomic_load(&sum->n_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.cand = cand;
                out.chunk_max = chunk_max;
                out.pad2 = 0;
                t.host->s = out;
            }
            // every lane's stores above precede this instruction in the wave's program order; the release waits for them all
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(&t.host->seq, t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <bool HI>
__global__ __launch_bounds__(kK1Threads) void k_union_tail(UnionTailArgs t) { union_tail_body<HI>(t, (int)blockIdx.x); }

template <int UNROLL, bool NT, class KT>
__global__ __launch_bounds__(kK1Threads) void k_scan_batch(BatchScanArgs<KT> a)
{
    scan_batch_body<UNROLL, NT, KT>(a, (int)blockIdx.x, (int)gridDim.x);
}

// table pass of batch i+1 with the union tail of batch i in its first blocks (see k_scan_keyed_with_tail)
template <int UNROLL, bool NT, class KT, bool HI>
__global__ __launch_bounds__(kK1Threads) void k_scan_batch_with_tail(BatchScanArgs<KT> a, UnionTailArgs t)
{
    const int n_tail = t.tiles;
    if ((int)blockIdx.x < n_tail) union_tail_body<HI>(t, (int)blockIdx.x);
    else scan_batch_body<UNROLL, NT, KT>(a, (int)blockIdx.x - n_tail, (int)gridDim.x - n_tail);
}

// ---- per-query results of a batch, materialised from its union on request (pie_batch_read_results / _result_device_ptrs, the
// per-query exchange messages): counts[U], offsets[U+1] and the row list of query q, bit for bit what a single scan of that
// query gives.  Three small launches for any number of queries (blockIdx.y = index into the list of queries asked for): per-tile
// counts, a one-block prefix of the tile sums per query, the write.  Off the hot path: no tickets, no spinning.
struct MatArgs {
    int n_users, tiles, n_list;
    const long long* uoff;
    const int* urows;
    const unsigned* umlo;
    const unsigned* umhi;      // nullptr for batches of <= 32 queries
    int* counts;               // + q * users_stride
    long long* offsets;        // + q * users_stride
    long long users_stride;
    int* out[kBatchMax];       // row list of the i-th query asked for
    long long out_cap[kBatchMax];
    long long* tile_sum;       // [n_list][tiles + 1] scratch: tile sums, then their exclusive prefix (entry `tiles` = M)
    unsigned int* qmax;        // [n_list] largest per-user count
    unsigned char q_of[kBatchMax]; // the queries asked for
};

__device__ __forceinline__ int mat_count_user(const MatArgs& a, int u, int q)
{
    const long long lo = a.uoff[u], hi = a.uoff[u + 1];
    const unsigned* m = (q >= 32) ? a.umhi : a.umlo;
    int c = 0;
    for (long long j = lo; j < hi; ++j) c += (int)((m[j] >> (q & 31)) & 1u);
    return c;
}

__global__ __launch_bounds__(256) void k_mat_count(MatArgs a)
{
    __shared__ int s_w[4];
    __shared__ unsigned s_m[4];
    const int li = (int)blockIdx.y, q = a.q_of[li];
    const int u = (int)(blockIdx.x * 256 + threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = u < a.n_users ? mat_count_user(a, u, q) : 0;
    if (u < a.n_users) a.counts[(long long)q * a.users_stride + u] = c;
    int v = c;
    unsigned mx = (unsigned)c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o, kWave);
        mx = max(mx, (unsigned)__shfl_xor((int)mx, o, kWave));
    }
    if (lane == 0) { s_w[wave] = v; s_m[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_sum[(long long)li * (a.tiles + 1) + blockIdx.x] = (long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const unsigned m4 = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m4) atomicMax(&a.qmax[li], m4);
    }
}

// one block per query: exclusive prefix of its tile sums in place; entry `tiles` receives M
__global__ __launch_bounds__(1024) void k_mat_prefix(MatArgs a)
{
    __shared__ long long s_w[16];
    __shared__ long long s_carry;
    long long* ts = a.tile_sum + (long long)blockIdx.x * (a.tiles + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < a.tiles; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const long long v = i < a.tiles ? ts[i] : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const long long tt = __shfl_up(inc, o, kWave);
            if (lane >= o) inc += tt;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        long long before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < a.tiles) ts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ts[a.tiles] = s_carry;
        a.offsets[(long long)a.q_of[blockIdx.x] * a.users_stride + a.n_users] = s_carry;
    }
}

__global__ __launch_bounds__(256) void k_mat_write(MatArgs a)
{
    __shared__ int s_w[4];
    const int li = (int)blockIdx.y, q = a.q_of[li];
    const int u = (int)(blockIdx.x * 256 + threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = u < a.n_users ? mat_count_user(a, u, q) : 0;
    const int inc = wave_incl_scan_i32(c, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    long long at = a.tile_sum[(long long)li * (a.tiles + 1) + blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) at += s_w[w];
    if (u >= a.n_users) return;
    a.offsets[(long long)q * a.users_stride + u] = at;
    if (c == 0 || at + c > a.out_cap[li]) return;
    int* out = a.out[li];
    const unsigned* m = (q >= 32) ? a.umhi : a.umlo;
    const long long lo = a.uoff[u], hi = a.uoff[u + 1];
    for (long long j = lo; j < hi; ++j)
        if ((m[j] >> (q & 31)) & 1u) out[at++] = a.urows[j];
}

// ---- the requests of a batch, fetched together (pie_batch_fetch_requests): request i = (query qi[i], user usr[i]); its feed is
// the rows of the user's union slice that carry the query's bit.  k_req_count: rows per request; after the prefix, k_req_write
// writes every request's rows and their columns (start, end, disc) behind off[i] — what the host serialises, in ONE round trip
// instead of three small copies per request.
__global__ __launch_bounds__(256) void k_req_count(int n_req, const int* __restrict__ qi, const int* __restrict__ usr, int n_users,
                                                   const long long* __restrict__ uoff, const unsigned* __restrict__ umlo,
                                                   const unsigned* __restrict__ umhi, int* __restrict__ cnt)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n_req) return;
    const int u = usr[i], q = qi[i];
    int c = 0;
    if ((unsigned)u < (unsigned)n_users) {
        const unsigned* m = (q >= 32) ? umhi : umlo;
        for (long long j = uoff[u]; j < uoff[u + 1]; ++j) c += (int)((m[j] >> (q & 31)) & 1u);
    }
    cnt[i] = c;
}
__global__ __launch_bounds__(256) void k_req_write(int n_req, const int* __restrict__ qi, const int* __restrict__ usr, int n_users,
                                                   const long long* __restrict__ uoff, const int* __restrict__ urows,
                                                   const unsigned* __restrict__ umlo, const unsigned* __restrict__ umhi,
                                                   const long long* __restrict__ off, const PayRec* __restrict__ pay,
                                                   const long long* __restrict__ end, long long cap, int* __restrict__ idx_out,
                                                   long long* __restrict__ start_out, long long* __restrict__ end_out, int* __restrict__ disc_out)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n_req) return;
    const int u = usr[i], q = qi[i];
    if ((unsigned)u >= (unsigned)n_users) return;
    const unsigned* m = (q >= 32) ? umhi : umlo;
    long long at = off[i];
    for (long long j = uoff[u]; j < uoff[u + 1]; ++j)
        if ((m[j] >> (q & 31)) & 1u) {
            if (at < cap) {
                const int r = urows[j];
                const PayRec p = pay[r];
                idx_out[at] = r;
                start_out[at] = p.start;
                end_out[at] = end[r];
                disc_out[at] = p.disc;
            }
            ++at;
        }
}

// the union arrays as one int32 message in caller-owned device-visible memory (layout: UnionTailArgs::msg)
__global__ __launch_bounds__(256) void k_union_pack(int n_users, int u_pad, const long long* __restrict__ uoff, const int* __restrict__ urows,
                                                    const unsigned* __restrict__ umlo, const unsigned* __restrict__ umhi, long long cap,
                                                    int* __restrict__ dst)
{
    const long long mu = uoff[n_users];
    const long long k = mu < cap ? mu : cap;
    const long long head = (long long)u_pad + 2;
    int* rows = dst + head;
    int* lo = rows + cap;
    int* hi = lo + cap;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < head; i += stride)
        dst[i] = i <= n_users ? (int)uoff[i] : (int)mu;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
        rows[i] = urows[i];
        lo[i] = (int)umlo[i];
        if (umhi) hi[i] = (int)umhi[i];
    }
}

// ------------------------------------------------------------------------------------------------ K3 scatter

// Records of K1 block b (blk_count[b] of them, in that block's private region) -> bucket slot
// offsets[user] + rank.  No atomics: the rank came back from K1's histogram atomic.  One block per region
// (grid-stride when there are more regions than blocks): the 256 threads take the region's records in one step
// in the common case, so the kernel is one dependent chain (count -> record -> offset -> store) deep.
__global__ __launch_bounds__(256) void k_scatter(const SelRec* __restrict__ sel, const int* __restrict__ sel_rank,
                                                 const int* __restrict__ blk_count, int nb, long long rows_per_block,
                                                 const long long* __restrict__ offsets,
                                                 BktRec* __restrict__ bkt, HotSet hot, const int* __restrict__ blk_hot_base)
{
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const long long base = (long long)b * rows_per_block;
        // the first 256 records are fetched without waiting for the count (the region is at least one block tile
        // long, so the addresses are valid; entries past the count are simply not used)
        const int cnt = blk_count[b];
        SelRec rec = sel[base + threadIdx.x];
        int rank = sel_rank[base + threadIdx.x];
        for (int i = threadIdx.x; i < cnt; i += 256) {
            if (i >= 256) {
                rec = sel[base + i];
                rank = sel_rank[base + i];
            }
            long long pos = offsets[rec.user];
            if (rank < 0) pos += blk_hot_base[(long long)b * kHotMax + hot_slot_of(hot, rec.user)] + (rank & 0x7FFFFFFF); // hot user: block base + rank in block
            else pos += rank;
            BktRec out;
            out.start = rec.start;
            out.idx = rec.idx;
            out.pad = 0;
            bkt[pos] = out;
        }
    }
}

// Buckets that outgrew their direct slots: the records that did fit (ranks below the capacity) move behind offsets[u] in
// bkt, where K3 puts the staged rest.  One wave per listed bucket, coalesced 16-byte copies.  (-1 entries: buckets with
// nothing in the slots — hot users, tables without slots.)
__global__ __launch_bounds__(256) void k_copy_direct(const int* __restrict__ over_list, const Summary* __restrict__ summary,
                                                     DirectSlots direct, const long long* __restrict__ offsets,
                                                     BktRec* __restrict__ bkt)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned n_over = summary->n_over;
    const int dcap = 1 << direct.shift;
    for (unsigned w = blockIdx.x * 4 + wave; w < n_over; w += gridDim.x * 4) {
        const int u = over_list[w];
        if (u < 0) continue;
        const BktRec* src = slots_of(direct, u);
        BktRec* dst = bkt + offsets[u];
        for (int i = lane; i < dcap; i += kWave) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------ K4 order

// Register sorting network (bitonic, fully unrolled so every index is a compile-time constant): NS slots,
// the first n hold the bucket, the rest are +inf padding.
template <int NS>
__device__ __forceinline__ void sort_bucket_regs(int n, const BktRec* __restrict__ src, int* __restrict__ dst)
{
    long long ks[NS];
    int ki[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const bool in = k < n;
        BktRec r;
        r.start = INT64_MAX;
        r.idx = INT32_MAX;
        if (in) r = src[k];
        ks[k] = r.start;
        ki[k] = r.idx;
    }
#pragma unroll
    for (int k = 2; k <= NS; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool lt = key_less(ks[l], ki[l], ks[i], ki[i]); // element l sorts before element i
                    const bool sw = up ? lt : !lt;
                    const long long s0 = sw ? ks[l] : ks[i], s1 = sw ? ks[i] : ks[l];
                    const int i0 = sw ? ki[l] : ki[i], i1 = sw ? ki[i] : ki[l];
                    ks[i] = s0; ks[l] = s1; ki[i] = i0; ki[l] = i1;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (k < n) dst[k] = ki[k];
}

// K4a: one thread per bucket of <= kTinyMax (16) rows: every load issued at once, sorting network in
// registers (8 slots for the common case, 16 otherwise).  Keys (start, idx) are unique.
// With direct slots the bucket is read from direct[u * kTinyMax ...] (src) and written to out_idx[offsets[u] ...]; a
// bucket that outgrew the slots has its first kTinyMax records copied behind offsets[u] in bkt, where K3 puts the rest.
__device__ __forceinline__ void sort_tiny_bucket(int u, const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                 BktRec* __restrict__ bkt, const DirectSlots& direct,
                                                 int* __restrict__ out_idx, const HotSet& hot)
{
    const int n = counts[u];
    if (n == 0 || n > kTinyMax) return; // larger buckets were listed by K2 for the wave / block sorts
    if (hot.n > 0 && hot_slot_of(hot, u) >= 0) return; // hot user: rows staged, bucket listed for K3 + K4 by K2
    const long long o = offsets[u];
    const BktRec* src = direct.p ? slots_of(direct, u) : bkt + o;
    if (n == 1) { out_idx[o] = src[0].idx; return; }
    if (n <= 8) sort_bucket_regs<8>(n, src, out_idx + o);
    else sort_bucket_regs<16>(n, src, out_idx + o);
}

// K4b: one block per segment (<= kSegMax rows): bitonic sort of (start, idx) in LDS.
// K4 (tiny buckets): one thread per user, buckets of <= 16 rows sorted in registers.
__global__ __launch_bounds__(256) void k_sort_tiny(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                   int n_users, BktRec* __restrict__ bkt, DirectSlots direct,
                                                   int* __restrict__ out_idx, HotSet hot)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u < n_users) sort_tiny_bucket(u, counts, offsets, bkt, direct, out_idx, hot);
}

// K4 (segments): buckets of 513..4096 rows, and the 4096-row tiles of bigger buckets: one 1024-thread block each.
// A bitonic sort whose 78 compare-exchange steps were all LDS + barrier in the first version; its latency (≈59 µs per
// 4096-row tile) is what a skewed table pays, since one hot bucket = one block.  Now every wave keeps its 256 elements
// in registers (four per lane): all steps whose partner lies inside the wave's chunk (j < 256: every step of the first
// eight stages, and the last eight steps of each later stage) run as lane shuffles / in-lane exchanges without a
// barrier, and only the j >= 256 steps of the last four stages go through LDS: 10 barrier steps + 8 hand-overs.
template <int EPL, int J>
__device__ __forceinline__ void sort2_inlane_step(long long (&ks)[EPL], int (&ki)[EPL], int lane, int k); // defined with the wave sort below

template <int EPL>
__device__ __forceinline__ void wave_stage(long long (&ks)[EPL], int (&ki)[EPL], int lane, int base, int k, int j_from)
{
    // steps j = j_from, j_from / 2, ..., 1 of bitonic stage k; element e of this lane has global index base + lane*EPL + e
    for (int j = j_from; j >= EPL; j >>= 1) { // partner in lane ^ (j / EPL), same slot
        const int lm = j / EPL;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int g = base + lane * EPL + e;
            const long long os = __shfl_xor(ks[e], lm, kWave);
            const int oi = __shfl_xor(ki[e], lm, kWave);
            const bool keep_min = ((g & j) == 0) == ((g & k) == 0);
            const bool other_less = key_less(os, oi, ks[e], ki[e]);
            const bool take = keep_min ? other_less : !other_less;
            ks[e] = take ? os : ks[e];
            ki[e] = take ? oi : ki[e];
        }
    }
    // partners inside the lane: the helper derives the direction from (lane_arg * EPL + e) & k, so it is handed the
    // lane's position in the whole segment (base is a multiple of EPL)
    const int seg_lane = base / EPL + lane;
    if constexpr (EPL >= 4) { if (j_from >= 2) sort2_inlane_step<EPL, 2>(ks, ki, seg_lane, k); }
    if constexpr (EPL >= 2) { if (j_from >= 1) sort2_inlane_step<EPL, 1>(ks, ki, seg_lane, k); }
}

__global__ __launch_bounds__(1024) void k_sort_segments(const Segment* __restrict__ seg_list, const Summary* __restrict__ summary,
                                                        BktRec* __restrict__ bkt, int* __restrict__ out_idx)
{
    constexpr int EPL = 4, kChunk = EPL * kWave; // 256 elements per wave
    __shared__ long long ks_s[kSegMax];
    __shared__ int ki_s[kSegMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = wave * kChunk;
    const unsigned n_seg = summary->n_seg;
    for (unsigned w = blockIdx.x; w < n_seg; w += gridDim.x) {
        const Segment sg = seg_list[w];
        int p = kChunk;
        while (p < sg.len) p <<= 1;
        const bool active = base < p; // waves beyond the padded length only keep the barriers company
        long long ks[EPL];
        int ki[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int g = base + lane * EPL + e;
            BktRec r;
            r.start = INT64_MAX;
            r.idx = INT32_MAX;
            if (active && g < sg.len) r = bkt[sg.pos + g];
            ks[e] = r.start;
            ki[e] = r.idx;
        }
        if (active)
            for (int k = 2; k <= kChunk; k <<= 1) wave_stage<EPL>(ks, ki, lane, base, k, k >> 1);
        for (int k = kChunk * 2; k <= p; k <<= 1) {
            if (active) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    ks_s[base + lane * EPL + e] = ks[e];
                    ki_s[base + lane * EPL + e] = ki[e];
                }
            }
            __syncthreads();
            for (int j = k >> 1; j >= kChunk; j >>= 1) {
                for (int t = threadIdx.x; t < (p >> 1); t += blockDim.x) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // index with bit j clear
                    const int l = i | j;
                    const bool up = (i & k) == 0;
                    const long long sa = ks_s[i], sb = ks_s[l];
                    const int ia = ki_s[i], ib = ki_s[l];
                    const bool swap = up ? key_less(sb, ib, sa, ia) : key_less(sa, ia, sb, ib);
                    if (swap) {
                        ks_s[i] = sb; ks_s[l] = sa;
                        ki_s[i] = ib; ki_s[l] = ia;
                    }
                }
                __syncthreads();
            }
            if (active) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    ks[e] = ks_s[base + lane * EPL + e];
                    ki[e] = ki_s[base + lane * EPL + e];
                }
                wave_stage<EPL>(ks, ki, lane, base, k, kChunk >> 1);
            }
            __syncthreads(); // the next stage's stores must not overtake another wave's loads of this stage
        }
        if (active) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int g = base + lane * EPL + e;
                if (g < sg.len) {
                    if (sg.flags & 1) {
                        BktRec r;
                        r.start = ks[e];
                        r.idx = ki[e];
                        r.pad = 0;
                        bkt[sg.pos + g] = r;
                    } else {
                        out_idx[sg.pos + g] = ki[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// K4 (small segments): buckets of 17..512 rows, ONE WAVE each, entirely in registers: lane l holds elements
// [l*EPL, (l+1)*EPL) of the padded bucket; bitonic steps with partner distance < EPL are compare-exchanges inside
// the lane, the others exchange with lane l ^ (j/EPL) through the cross-lane network (__shfl_xor).  No LDS storage,
// no barriers: the first version of this kernel kept the bucket in LDS and was LDS-bandwidth bound (0.92 ms for
// 10^5 buckets of ~254 rows); every loop below has compile-time bounds so all indices are static registers.
template <int EPL, int J>
__device__ __forceinline__ void sort2_inlane_step(long long (&ks)[EPL], int (&ki)[EPL], int lane, int k)
{
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int l = e ^ J;
        if (l > e) {
            const bool up = ((lane * EPL + e) & k) == 0;
            const bool lt = key_less(ks[l], ki[l], ks[e], ki[e]); // slot l sorts before slot e
            const bool sw = up ? lt : !lt;
            const long long s0 = sw ? ks[l] : ks[e], s1 = sw ? ks[e] : ks[l];
            const int i0 = sw ? ki[l] : ki[e], i1 = sw ? ki[e] : ki[l];
            ks[e] = s0; ks[l] = s1; ki[e] = i0; ki[l] = i1;
        }
    }
}

template <int EPL>
__device__ __forceinline__ void wave_sort_segment(const Segment sg, const BktRec* __restrict__ src, int* __restrict__ out_idx,
                                                  int lane)
{
    constexpr int P = EPL * 64;
    long long ks[EPL];
    int ki[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = lane * EPL + e;
        BktRec r;
        r.start = INT64_MAX;
        r.idx = INT32_MAX;
        if (i < sg.len) r = src[i];
        ks[e] = r.start;
        ki[e] = r.idx;
    }
    // real loops over the stages (see wave_sort3): the body stays resident in the instruction cache
#pragma nounroll
    for (int k = 2; k <= P; k <<= 1) {
#pragma nounroll
        for (int j = k >> 1; j >= EPL; j >>= 1) { // partner lives in lane ^ (j / EPL), same slot
            const int lm = j / EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = lane * EPL + e;
                const long long os = __shfl_xor(ks[e], lm, kWave);
                const int oi = __shfl_xor(ki[e], lm, kWave);
                const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                const bool other_less = key_less(os, oi, ks[e], ki[e]);
                const bool take = keep_min ? other_less : !other_less;
                ks[e] = take ? os : ks[e];
                ki[e] = take ? oi : ki[e];
            }
        }
        const int jmax = (k >> 1) < EPL ? (k >> 1) : (EPL >> 1);
        if constexpr (EPL >= 8) { if (jmax >= 4) sort2_inlane_step<EPL, 4>(ks, ki, lane, k); }
        if constexpr (EPL >= 4) { if (jmax >= 2) sort2_inlane_step<EPL, 2>(ks, ki, lane, k); }
        if constexpr (EPL >= 2) { if (jmax >= 1) sort2_inlane_step<EPL, 1>(ks, ki, lane, k); }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = lane * EPL + e;
        if (i < sg.len) out_idx[sg.pos + i] = ki[e];
    }
}

__global__ __launch_bounds__(256) void k_sort_small(const Segment* __restrict__ small_list, const Summary* __restrict__ summary,
                                                    const BktRec* __restrict__ bkt, DirectSlots direct, int* __restrict__ out_idx)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned n_small = summary->n_small;
    for (unsigned w = blockIdx.x * 4 + wave; w < n_small; w += gridDim.x * 4) {
        const Segment sg = small_list[w];
        // flags bit 1: the whole bucket still sits in its user's direct slots (user id in the bits above); else behind pos in bkt
        const BktRec* src = (sg.flags & 2) ? slots_of(direct, sg.flags >> 2) : bkt + sg.pos;
        if (sg.len <= 128) wave_sort_segment<2>(sg, src, out_idx, lane);
        else if (sg.len <= 256) wave_sort_segment<4>(sg, src, out_idx, lane);
        else wave_sort_segment<8>(sg, src, out_idx, lane);
    }
}

// K4c: one merge pass over every big bucket: sorted runs of `width` -> sorted runs of ways*width, ways <= kMergeWays.
// Each thread owns one input element and binary-searches its rank in the sibling runs of its group; its place in the
// merged group = its place in its own run + the number of smaller keys in every other run (keys are unique, so ranks are
// a permutation).  The searches of one element advance in LOCKSTEP — one probe per sibling run and step, all independent
// loads — so a 16-way pass costs about one search's latency chain (log2(width) dependent L2 loads), not sixteen: a bucket
// of up to 16 x 4096 rows (a Zipf head user) is merged by ONE launch instead of two 4-way ones.  grid.y indexes big_list.
constexpr int kMergeWays = 16;
__global__ __launch_bounds__(256) void k_merge_pass(const int* __restrict__ big_list, int n_big, const int* __restrict__ counts,
                                                    const long long* __restrict__ offsets, long long width, int ways,
                                                    const BktRec* __restrict__ src, BktRec* __restrict__ dst,
                                                    int* __restrict__ dst_idx_only)
{
    for (int bb = blockIdx.y; bb < n_big; bb += gridDim.y) {
        const int u = big_list[bb];
        const long long n = counts[u], o = offsets[u];
        for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
            const long long run = p / width;
            const long long group0 = (run / ways) * ways * width; // first slot of this group of `ways` runs
            const int my_r = (int)(run - (run / ways) * ways);
            const BktRec me = src[o + p];
            int lo[kMergeWays], hi[kMergeWays];
#pragma unroll
            for (int r = 0; r < kMergeWays; ++r) {
                const long long r0 = group0 + (long long)r * width;
                long long rn = n - r0;
                rn = rn > width ? width : rn;
                lo[r] = 0;
                hi[r] = (r < ways && r != my_r && rn > 0) ? (int)rn : 0; // an empty range: nothing to search
            }
            bool more = true;
            while (more) {
                more = false;
                BktRec x[kMergeWays];
#pragma unroll
                for (int r = 0; r < kMergeWays; ++r)
                    if (lo[r] < hi[r]) x[r] = src[o + group0 + (long long)r * width + ((lo[r] + hi[r]) >> 1)];
#pragma unroll
                for (int r = 0; r < kMergeWays; ++r) {
                    if (lo[r] < hi[r]) {
                        const int mid = (lo[r] + hi[r]) >> 1;
                        if (key_less(x[r].start, x[r].idx, me.start, me.idx)) lo[r] = mid + 1; else hi[r] = mid;
                        more |= lo[r] < hi[r];
                    }
                }
            }
            long long smaller = 0;
#pragma unroll
            for (int r = 0; r < kMergeWays; ++r) smaller += lo[r];
            const long long q = o + group0 + (p - run * width) + smaller;
            if (dst_idx_only) dst_idx_only[q] = me.idx; // last pass: only the row order is wanted
            else dst[q] = me;
        }
    }
}

// ------------------------------------------------------------------------------------------------ result packing

// One contiguous int32 message for the multi-GPU exchange:
//   [ off[0..u_pad] (u_pad+1 entries: exclusive offsets, off[u] = M for u >= U) | M | rows[0..min(M,cap)) ]
// Offsets instead of counts: the receiver slices Feed(rank, u) = rows[off[u] : off[u+1]] with no prefix sum of its
// own.  One launch instead of three D2D copies; the message feeds a single RCCL all-gather.
__global__ __launch_bounds__(256) void k_pack_results(const long long* __restrict__ offsets, int n_users, int u_pad,
                                                      const Summary* __restrict__ summary, const int* __restrict__ out_idx,
                                                      long long cap, int* __restrict__ dst)
{
    const long long m = (long long)summary->m;
    const long long k = m < cap ? m : cap;
    const long long head = (long long)u_pad + 2;
    const long long total = head + k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int v;
        if (i <= n_users) v = (int)offsets[i];
        else if (i <= u_pad) v = (int)m;
        else if (i == u_pad + 1) v = (int)m;
        else v = out_idx[i - head];
        dst[i] = v;
    }
}

// the same message from lists whose M the host knows (a batch's per-query lists)
__global__ __launch_bounds__(256) void k_pack_lists(const long long* __restrict__ offsets, int n_users, int u_pad, long long m,
                                                    const int* __restrict__ out_idx, long long cap, int* __restrict__ dst)
{
    const long long k = m < cap ? m : cap;
    const long long head = (long long)u_pad + 2;
    const long long total = head + k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int v;
        if (i <= n_users) v = (int)offsets[i];
        else if (i < head) v = (int)m;
        else v = out_idx[i - head];
        dst[i] = v;
    }
}

// The UNION message of a batch (multi-GPU exchange).  The Q queries of a batch are requests of the same few seconds: their
// row lists are almost the same rows, Q times over.  Per user the union of the Q lists, in (start, row) order, with a query
// mask per row, is what the exchange needs to move — 8 B per union row instead of 4 B per row per query:
//   [ uoff[0..u_pad] (exclusive offsets into the union rows, = Mu for u >= U) | Mu | rows[0..cap) | masks[0..cap) ]   int32 words
// Feed(q, u) = the rows of rows[uoff[u] : uoff[u+1]] whose mask has bit q, in that order.  Built from the finished per-query
// lists, whichever path produced them: one thread per user merges its (short) lists; a user with more than kUnionMax (32) union
// rows makes the message unusable (Mu = -1: the caller falls back to the per-query messages).
constexpr int kUnionMax = 32;
constexpr int kUnionThreads = 128;
struct UnionLists {
    const int* idx[kBatchMax]; // row list of query q (device)
};
struct alignas(8) UnionRow {
    int row;
    unsigned mask;
};

// one thread per user; its working list lives in LDS, slot-major ([slot][thread]: neighbouring threads touch neighbouring
// words), so the duplicate search and the insertion sort index it freely without touching global memory
__global__ __launch_bounds__(kUnionThreads) void k_union_collect(int n_q, int n_users, const long long* __restrict__ offsets,
                                                                 long long users_stride, UnionLists lists,
                                                                 const long long* __restrict__ start, UnionRow* __restrict__ scratch,
                                                                 int* __restrict__ ucnt, int* __restrict__ over)
{
    __shared__ int l_row[kUnionMax][kUnionThreads];
    __shared__ unsigned l_mask[kUnionMax][kUnionThreads];
    __shared__ long long l_start[kUnionMax][kUnionThreads];
    const int t = threadIdx.x;
    const int u = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (u >= n_users) return;
    int n = 0;
    bool overflow = false;
    for (int q = 0; q < n_q; ++q) {
        const long long lo = offsets[(long long)q * users_stride + u], hi = offsets[(long long)q * users_stride + u + 1];
        for (long long j = lo; j < hi; ++j) {
            const int r = lists.idx[q][j];
            int at = -1;
            for (int i = 0; i < n; ++i)
                if (l_row[i][t] == r) at = i;
            if (at >= 0) l_mask[at][t] |= 1u << q;
            else if (n < kUnionMax) {
                l_row[n][t] = r;
                l_mask[n][t] = 1u << q;
                l_start[n][t] = start[r];
                ++n;
            } else overflow = true;
        }
    }
    // (start, row) order: insertion sort of a handful of rows
    for (int i = 1; i < n; ++i) {
        const int xr = l_row[i][t];
        const unsigned xm = l_mask[i][t];
        const long long xs = l_start[i][t];
        int j = i - 1;
        while (j >= 0) {
            const long long ys = l_start[j][t];
            const int yr = l_row[j][t];
            if (ys < xs || (ys == xs && yr < xr)) break;
            l_row[j + 1][t] = yr;
            l_mask[j + 1][t] = l_mask[j][t];
            l_start[j + 1][t] = ys;
            --j;
        }
        l_row[j + 1][t] = xr;
        l_mask[j + 1][t] = xm;
        l_start[j + 1][t] = xs;
    }
    UnionRow* mine = scratch + (long long)u * kUnionMax;
    for (int i = 0; i < n; ++i) {
        UnionRow x;
        x.row = l_row[i][t];
        x.mask = l_mask[i][t];
        mine[i] = x;
    }
    ucnt[u] = n;
    if (overflow) atomicOr(over, 1);
}

// uoff[u] = group_base[u >> 10] + unit_local[u]: the two-level prefix of the union counts (k_ord_prefix,
// the kernels the ordered run uses for its unit counts)
__global__ __launch_bounds__(256) void k_union_write(int n_users, int u_pad, const int* __restrict__ unit_local,
                                                     const long long* __restrict__ group_base, const int* __restrict__ ucnt,
                                                     const UnionRow* __restrict__ scratch, const int* __restrict__ over, long long cap,
                                                     int* __restrict__ dst)
{
    const long long n_groups = ((long long)n_users + 1023) >> 10;
    const long long mu = group_base[n_groups];
    const int bad = *over;
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u <= (long long)u_pad + 1; u += (long long)gridDim.x * blockDim.x) {
        if (u < n_users) dst[u] = (int)(group_base[u >> 10] + unit_local[u]);
        else if (u <= u_pad) dst[u] = (int)mu;
        else dst[u] = bad ? -1 : (int)mu; // the word behind the offsets: Mu, or -1 when a user's union outgrew kUnionMax
    }
    int* rows = dst + u_pad + 2;
    int* masks = rows + cap;
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_users; u += (long long)gridDim.x * blockDim.x) {
        const long long at = group_base[u >> 10] + unit_local[u];
        const int n = ucnt[u];
        const UnionRow* mine = scratch + u * kUnionMax;
        for (int i = 0; i < n; ++i)
            if (at + i < cap) {
                rows[at + i] = mine[i].row;
                masks[at + i] = (int)mine[i].mask;
            }
    }
}

// ------------------------------------------------------------------------------------------------ table maintenance

__global__ __launch_bounds__(256) void k_set_end(long long* __restrict__ end, const int* __restrict__ rows,
                                                 const long long* __restrict__ new_end, long long k, long long n,
                                                 lkey_t* __restrict__ key, long long key_base, int key_shift,
                                                 fkey_t* __restrict__ fkey, long long fkey_base, int fkey_shift, OrdMirror ord)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k && (unsigned)rows[t] < (unsigned long long)n) {
        end[rows[t]] = new_end[t];
        const unsigned kk = key_of(new_end[t], key_base, key_shift), fk = key_of(new_end[t], fkey_base, fkey_shift, kFineKeyMax);
        if (key) key[rows[t]] = (lkey_t)kk;
        if (fkey) fkey[rows[t]] = (fkey_t)fk;
        ord_mirror_end(ord, rows[t], new_end[t], kk, fk);
    }
}

// createSession, k rows at a time (pie_append_rows' in-place path): the packed staging block [start k | end k | user k |
// disc k] becomes rows [row0, row0 + k) of the four columns and of the derived columns (both keys under the table's current
// parameters, the payload record), and user ids outside [0, n_users) are counted — one kernel instead of four copies, a
// validation pass and two key passes.
__global__ __launch_bounds__(256) void k_append_rows(const long long* __restrict__ st_start, const long long* __restrict__ st_end,
                                                     const int* __restrict__ st_user, const int* __restrict__ st_disc, long long k,
                                                     long long row0, int n_users, long long* __restrict__ start,
                                                     long long* __restrict__ end, int* __restrict__ user, int* __restrict__ disc,
                                                     lkey_t* __restrict__ key, long long key_base, int key_shift,
                                                     fkey_t* __restrict__ fkey, long long fkey_base, int fkey_shift,
                                                     PayRec* __restrict__ pay, unsigned int* __restrict__ bad)
{
    unsigned int local = 0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < k; t += (long long)gridDim.x * blockDim.x) {
        const long long r = row0 + t;
        const long long sv = st_start[t], ev = st_end[t];
        const int uv = st_user[t], dv = st_disc[t];
        local += ((unsigned)uv >= (unsigned)n_users) ? 1u : 0u;
        start[r] = sv;
        end[r] = ev;
        user[r] = uv;
        disc[r] = dv;
        if (key) key[r] = (lkey_t)key_of(ev, key_base, key_shift);
        if (fkey) fkey[r] = (fkey_t)key_of(ev, fkey_base, fkey_shift, kFineKeyMax);
        if (pay) {
            PayRec pr;
            pr.start = sv;
            pr.user = uv;
            pr.disc = dv;
            pay[r] = pr;
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, kWave);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

__global__ __launch_bounds__(256) void k_fetch_rows(const int* __restrict__ idx, long long m, long long n,
                                                    const long long* __restrict__ start, const long long* __restrict__ end,
                                                    const int* __restrict__ user, const int* __restrict__ disc,
                                                    long long* __restrict__ o_start, long long* __restrict__ o_end,
                                                    int* __restrict__ o_user, int* __restrict__ o_disc)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const long long r = idx[t];
    const bool ok = r >= 0 && r < n;
    o_start[t] = ok ? start[r] : 0;
    o_end[t] = ok ? end[r] : INT64_MIN;
    o_user[t] = ok ? user[r] : -1;
    o_disc[t] = ok ? disc[r] : -1;
}

__global__ __launch_bounds__(256) void k_validate_users(const int* __restrict__ user, long long n, int n_users,
                                                        unsigned int* __restrict__ bad)
{
    unsigned int local = 0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x)
        local += ((unsigned)user[r] >= (unsigned)n_users) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, kWave);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

// ------------------------------------------------------------------------------------------------ ordered row lists

// Order-preserving compaction of the rows matching a one-column predicate, in three steps:
// per-block counts -> prefix (k_block_prefix) -> ordered write.
//   MODE 0  newly expired: prev_now < end <= now ("next" row of SURVEY.md §8f-1: dead per
//           /root/reference/server/sessionStore.js:69 at `now`, not yet dead at `prev_now`)
//   MODE 1  deleteSessionsForUser (/root/reference/server/sessionStore.js:55-64): user == target, strict
//           match; the write step tombstones the row (end = INT64_MIN: never live again)
//   MODE 3  retention purge with calendar months (see add_months_ms); tombstones like MODE 1; `aux` = start column
//   MODE 2  _pruneCalendarEvents (/root/reference/server/storage/sqlProvider.js:956-968): start < cutoff, the
//           complement of the window predicate; tombstones like MODE 1.  `aux` is the start column here.
// Calendar-month arithmetic of /root/reference/server/storage/sqlProvider.js:999-1009 (_addMonths:
// `date.setMonth(date.getMonth() + months)` on a local-time Date), in integers: local = UTC + tz (fixed offset),
// civil date from the day number, month index shifted, day number of (year', month', 1) + (day - 1) — the
// normalisation JS's MakeDay performs, so "Dec 31 + 2 months" lands on Mar 3 (Mar 2 in a leap year) — and back.
// JS Date range rules: |ts| > 8.64e15 is an invalid Date and comes back unchanged; a result outside the range is
// NaN (ok = false), for which `now >= expiry` is false.
__device__ __forceinline__ long long floor_div(long long a, long long b) { const long long q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

// A zone as the host's transition table in device memory (pie_retention_purge_tz): words [0] = n, [1] = months, then T[n] (UTC
// instants of the transitions), L[n] (the same instants on the OLD offset's local clock: T[i] + off[i]), off[n + 1]
// (off[0] before T[0], off[i + 1] from T[i] on).  ECMA-262 LocalTime / UTC: a local time that is skipped or repeated at a
// transition is read with the offset before the transition.
struct TzView {
    const long long* T;
    const long long* L;
    const long long* off;
    int n;
};
__device__ __forceinline__ TzView tz_view(const long long* p)
{
    TzView z;
    z.n = (int)p[0];
    z.T = p + 2;
    z.L = z.T + z.n;
    z.off = z.L + z.n;
    return z;
}
// entries of an ascending table that are <= x
__device__ __forceinline__ int tz_rank(const long long* tab, int n, long long x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ long long tz_utc_from_local(const TzView& z, long long tl)
{
    const int k = tz_rank(z.L, z.n, tl);
    long long u = tl - z.off[k];
    if (k >= 1 && u < z.T[k - 1]) u = tl - z.off[k - 1]; // skipped local time: the offset before the transition
    return u;
}

// the month shift on the LOCAL fields: local -> local (JS MakeDay normalisation: day overflow rolls into the next month)
__device__ __forceinline__ long long shift_months_local(long long local, int months);

__device__ __forceinline__ long long add_months_ms(long long ts, int months, long long tz_ms, bool& ok)
{
    constexpr long long kMax = 8640000000000000LL;
    ok = true;
    if (ts > kMax || ts < -kMax) return ts;
    const long long out = shift_months_local(ts + tz_ms, months) - tz_ms; // |ts| <= 8.64e15, |tz| <= a day: no overflow
    if (out > kMax || out < -kMax) ok = false;
    return out;
}

__device__ __forceinline__ long long add_months_tz(long long ts, int months, const TzView& z, bool& ok)
{
    constexpr long long kMax = 8640000000000000LL;
    ok = true;
    if (ts > kMax || ts < -kMax) return ts;
    const long long local = ts + z.off[tz_rank(z.T, z.n, ts)];
    const long long out = tz_utc_from_local(z, shift_months_local(local, months));
    if (out > kMax || out < -kMax) ok = false;
    return out;
}

__device__ __forceinline__ long long shift_months_local(long long local, int months)
{
    constexpr long long kDay = 86400000LL;
    const long long days = floor_div(local, kDay);
    const long long ms_of_day = local - days * kDay;
    // civil_from_days (proleptic Gregorian; day 0 = 1970-01-01)
    long long z = days + 719468;
    const long long era = floor_div(z, 146097);
    const long long doe = z - era * 146097;                                  // [0, 146096]
    const long long yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365; // [0, 399]
    const long long doy = doe - (365 * yoe + yoe / 4 - yoe / 100);            // [0, 365]
    const long long mp = (5 * doy + 2) / 153;                                 // [0, 11], March-based
    const long long d = doy - (153 * mp + 2) / 5 + 1;                         // [1, 31]
    const long long m = mp < 10 ? mp + 3 : mp - 9;                            // [1, 12]
    const long long y = yoe + era * 400 + (m <= 2 ? 1 : 0);
    // shift the month index, renormalise the year
    const long long mi = (m - 1) + months;
    const long long y2 = y + floor_div(mi, 12);
    const long long m2 = mi - floor_div(mi, 12) * 12 + 1;                     // [1, 12]
    // days_from_civil(y2, m2, 1) + (d - 1)
    const long long yy = y2 - (m2 <= 2 ? 1 : 0);
    const long long era2 = floor_div(yy, 400);
    const long long yoe2 = yy - era2 * 400;
    const long long doy2 = (153 * (m2 > 2 ? m2 - 3 : m2 + 9) + 2) / 5;        // day of (March-based) year of the 1st
    const long long doe2 = yoe2 * 365 + yoe2 / 4 - yoe2 / 100 + doy2;
    const long long days2 = era2 * 146097 + doe2 - 719468 + (d - 1);
    return days2 * kDay + ms_of_day;                                          // < ~9e15 * small: fits
}

template <int MODE>
__device__ __forceinline__ bool list_match(const long long* __restrict__ end, const int* __restrict__ user, long long r,
                                           long long a, long long b)
{
    if constexpr (MODE == 0) {
        const long long e = end[r];
        return e <= b && e > a;
    } else if constexpr (MODE == 1) {
        return user[r] == (int)a && end[r] != INT64_MIN;
    } else if constexpr (MODE == 2) {
        return reinterpret_cast<const long long*>(user)[r] < a && end[r] != INT64_MIN;
    } else {
        // MODE 3  retention purge (/root/reference/server/storage/sqlProvider.js:863-890,991-997): now >= addMonths(start,
        //         months); `a` = now, `b` packs months (low 16 bits, signed) and the zone offset in minutes (above)
        if (end[r] == INT64_MIN) return false;
        bool ok;
        if constexpr (MODE == 4) { // ... under a real time zone: `b` = device address of the zone's transition table (TzView)
            const long long* tab = reinterpret_cast<const long long*>(b);
            const long long expiry = add_months_tz(reinterpret_cast<const long long*>(user)[r], (int)tab[1], tz_view(tab), ok);
            return ok && a >= expiry;
        }
        const int months = (int)(short)(b & 0xFFFF);
        const long long tz_ms = (b >> 16) * 60000LL;
        const long long expiry = add_months_ms(reinterpret_cast<const long long*>(user)[r], months, tz_ms, ok);
        return ok && a >= expiry;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_list_count(const long long* __restrict__ end, const int* __restrict__ user, long long n,
                                                    long long rows_per_block, long long a, long long b,
                                                    int* __restrict__ blk_count)
{
    __shared__ long long lds4[4];
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    const long long c1 = min(n, c0 + rows_per_block);
    long long local = 0;
    for (long long r = c0 + threadIdx.x; r < c1; r += blockDim.x) local += list_match<MODE>(end, user, r, a, b) ? 1 : 0;
    local = block_sum_256(local, lds4);
    if (threadIdx.x == 0) blk_count[blockIdx.x] = (int)local;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_list_write(long long* __restrict__ end, const int* __restrict__ user, long long n,
                                                    long long rows_per_block, long long a, long long b,
                                                    const long long* __restrict__ blk_off, int* __restrict__ queue, long long cap,
                                                    lkey_t* __restrict__ key, fkey_t* __restrict__ fkey, OrdMirror ord)
{
    __shared__ int wcount[4];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    const long long c1 = min(n, c0 + rows_per_block);
    if (threadIdx.x == 0) carry_s = blk_off[blockIdx.x];
    __syncthreads();
    for (long long r0 = c0; r0 < c1; r0 += blockDim.x) {
        const long long r = r0 + threadIdx.x;
        const bool hit = r < c1 && list_match<MODE>(end, user, r, a, b);
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        long long base = carry_s;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        const long long pos = base + prefix_in_ballot(bal);
        if (hit && pos < cap) queue[pos] = (int)r;
        if constexpr (MODE != 0) {
            if (hit) { // a tombstone's liveness keys are 0 under every base
                end[r] = INT64_MIN;
                if (key) key[r] = 0;
                if (fkey) fkey[r] = 0;
                ord_mirror_end(ord, r, INT64_MIN, 0u, 0u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ user-hash sharding on the device
//
// rows are partitioned by gpu = splitmix64(user) mod G (pie_shard_of; the same function in the oracle).
// A rank that holds the whole table keeps the rows of its own users, in table order, with the users re-numbered densely
// in ascending global id — all on the device: user flags -> prefix (= the local ids), per-block row counts -> prefix ->
// order-preserving compaction of the four columns.  The maps back (local row -> global row, local user -> global user)
// stay on the device for pie_shard_maps.
__device__ __forceinline__ int shard_of_user(int user, int n_shards)
{
    unsigned long long z = (unsigned long long)(unsigned int)user + 0x9E3779B97F4A7C15ULL;
    return (int)(mix64(z) % (unsigned long long)n_shards);
}

__global__ __launch_bounds__(256) void k_shard_user_flags(int n_users, int rank, int world, int* __restrict__ flag)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u < n_users) flag[u] = shard_of_user(u, world) == rank ? 1 : 0;
}

// local_of_global[u] = dense local id (exclusive prefix of the flags) or -1; users_global[local id] = u
__global__ __launch_bounds__(256) void k_shard_user_ids(int n_users, const int* __restrict__ flag, const long long* __restrict__ off,
                                                        int* __restrict__ local_of_global, int* __restrict__ users_global)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const int l = (int)off[u];
    local_of_global[u] = flag[u] ? l : -1;
    if (flag[u]) users_global[l] = u;
}

__global__ __launch_bounds__(256) void k_shard_row_count(const int* __restrict__ user, long long n, long long rows_per_block,
                                                         const int* __restrict__ local_of_global, int n_users, int* __restrict__ blk_count)
{
    __shared__ long long lds4[4];
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    const long long c1 = min(n, c0 + rows_per_block);
    long long local = 0;
    for (long long r = c0 + threadIdx.x; r < c1; r += blockDim.x) {
        const int u = user[r];
        local += ((unsigned)u < (unsigned)n_users && local_of_global[u] >= 0) ? 1 : 0;
    }
    local = block_sum_256(local, lds4);
    if (threadIdx.x == 0) blk_count[blockIdx.x] = (int)local;
}

__global__ __launch_bounds__(256) void k_shard_row_write(const long long* __restrict__ start, const long long* __restrict__ end,
                                                         const int* __restrict__ user, const int* __restrict__ disc, long long n,
                                                         long long rows_per_block, const int* __restrict__ local_of_global, int n_users,
                                                         const long long* __restrict__ blk_off, long long* __restrict__ o_start,
                                                         long long* __restrict__ o_end, int* __restrict__ o_user, int* __restrict__ o_disc,
                                                         int* __restrict__ o_row)
{
    __shared__ int wcount[4];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    const long long c1 = min(n, c0 + rows_per_block);
    if (threadIdx.x == 0) carry_s = blk_off[blockIdx.x];
    __syncthreads();
    for (long long r0 = c0; r0 < c1; r0 += blockDim.x) {
        const long long r = r0 + threadIdx.x;
        int lu = -1;
        if (r < c1) {
            const int u = user[r];
            if ((unsigned)u < (unsigned)n_users) lu = local_of_global[u];
        }
        const bool hit = lu >= 0;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        long long base = carry_s;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        const long long pos = base + prefix_in_ballot(bal);
        if (hit) {
            o_start[pos] = start[r];
            o_end[pos] = end[r];
            o_user[pos] = lu;
            o_disc[pos] = disc[r];
            o_row[pos] = (int)r;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
}

// exclusive prefix over blk[0..nb) by ONE block of 1024 threads in a single sweep: thread t owns the contiguous
// chunk [t*per, (t+1)*per), so 16 K wave counts cost one block-wide scan instead of 64 of them
// host != nullptr: the total also goes to mapped host memory (seq last) — the caller learns the queue's length when THIS kernel
// ends, not after a copy behind the gather that follows it
__global__ __launch_bounds__(1024) void k_block_prefix_wide(const int* __restrict__ blk, int nb, long long* __restrict__ off,
                                                            unsigned long long* __restrict__ total_out, HostSummary* __restrict__ host,
                                                            unsigned long long seq)
{
    __shared__ long long wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (nb + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    long long mine = 0;
    for (int k = 0; k < per; ++k)
        if (b0 + k < nb) mine += blk[b0 + k];
    const long long incl = wave_incl_scan(mine, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int k = 0; k < per; ++k) {
        if (b0 + k < nb) {
            off[b0 + k] = run;
            run += blk[b0 + k];
        }
    }
    if (threadIdx.x == 1023) {
        off[nb] = run;
        if (total_out) *total_out = (unsigned long long)run;
        if (host) {
            host->s.m = (unsigned long long)run;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------------ expired queue, one pass

// The newly-expired change predicate (prev_now < end <= now) reads ONE column, so its roofline is 8 B/row.  The
// count / prefix / write form above reads it twice; this form reads it once: every WAVE owns a contiguous row
// range and the same index range of `stage` as a private, order-preserving output region (ballot + mbcnt prefix,
// direct stores, no LDS), a single block turns the per-wave counts into offsets, and a gather moves the few hits.
// Order of the final queue = ascending row index (the sequential-await order of
// /root/reference/server/storage/sqlProvider.js:834-861).
template <int UNROLL>
__global__ __launch_bounds__(kK1Threads) void k_expired_stage(const long long* __restrict__ end, long long n,
                                                              long long rows_per_block, long long prev_now, long long now,
                                                              int* __restrict__ stage, int* __restrict__ wave_count)
{
    constexpr int kTile = kUnitRows * UNROLL;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const long long rows_per_wave = rows_per_block / kK1Waves; // a multiple of every tile size (block range is 4096-aligned)
    const long long w0 = (long long)blockIdx.x * rows_per_block + (long long)wave * rows_per_wave;
    long long w1 = w0 + rows_per_wave;
    if (w1 > n) w1 = n;
    int fill = 0; // wave-uniform
    int* out = stage + w0;
    for (long long t = w0; t < w1; t += kTile) {
        if (t + kTile <= w1) {
            ll2_t e[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j)
                e[j] = stream_load<true>(reinterpret_cast<const ll2_t*>(end + t + j * kUnitRows + 2 * lane));
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const bool h0 = (e[j].x <= now) & (e[j].x > prev_now);
                const bool h1 = (e[j].y <= now) & (e[j].y > prev_now);
                const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1);
                if ((b0 | b1) == 0) continue;
                const int r = (int)(t + j * kUnitRows + 2 * lane);
                const int pos = fill + prefix_in_ballot(b0) + prefix_in_ballot(b1); // rows 2l, 2l+1 are adjacent
                if (h0) out[pos] = r;
                if (h1) out[pos + (h0 ? 1 : 0)] = r + 1;
                fill += __popcll(b0) + __popcll(b1);
            }
        } else {
            for (long long r0 = t; r0 < w1; r0 += kWave) {
                const long long r = r0 + lane;
                bool h = false;
                if (r < w1) {
                    const long long ev = end[r];
                    h = (ev <= now) & (ev > prev_now);
                }
                const unsigned long long b = __ballot(h);
                if (h) out[fill + prefix_in_ballot(b)] = (int)r;
                fill += __popcll(b);
            }
        }
    }
    if (lane == 0) wave_count[blockIdx.x * kK1Waves + wave] = w0 < n ? fill : 0;
}

// The same stage on the 2-byte liveness key (pie_kernels.h, "liveness-key column"): with kp = key(prev_now) and
// kn = key(now), a row with kp < key < kn is a hit without looking at `end` (key(end) > key(prev) => end > prev,
// key(end) < key(now) => end < now), a row with key == kp or key == kn needs the full compare, every other row is a
// miss.  2 B/row instead of 8 B/row; output identical (each lane holds eight consecutive rows, so lane order is row order).
// (the stage of block `bid`; returns the wave's number of hits, staged in order at stage[w0 ..))
template <int UNROLL>
__device__ __forceinline__ int expired_stage_keyed_body(const lkey_t* __restrict__ key, const long long* __restrict__ end, long long n,
                                                        long long rows_per_block, long long prev_now, long long now, unsigned kp, unsigned kn,
                                                        int* __restrict__ stage, int bid)
{
    constexpr int kTile = kKeyRowsPerLoad * UNROLL;
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const long long rows_per_wave = rows_per_block / kK1Waves; // a multiple of kTile (block range is 4096-aligned, UNROLL <= 2)
    const long long w0 = (long long)bid * rows_per_block + (long long)wave * rows_per_wave;
    long long w1 = w0 + rows_per_wave;
    if (w1 > n) w1 = n;
    int fill = 0; // wave-uniform
    int* out = stage + w0;
    const unsigned kp2 = kp | (kp << 16), kn2 = (kn | (kn << 16)) | 0x80008000u;
    auto is_hit = [&](unsigned k, long long r) -> bool {
        if (k > kp && k < kn) return true;
        if (k != kp && k != kn) return false;
        const long long ev = end[r];
        return (ev <= now) & (ev > prev_now);
    };
    for (long long t = w0; t < w1; t += kTile) {
        if (t + kTile <= w1) {
            u4_t kv[UNROLL];
#pragma unroll
            for (int j = 0; j < UNROLL; ++j)
                kv[j] = stream_load<true>(reinterpret_cast<const u4_t*>(key + t + j * kKeyRowsPerLoad + 8 * lane));
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                // candidates: kp <= key <= kn, per 16-bit half (keys < 2^15: neither subtraction borrows across halves)
                const unsigned w[4] = {kv[j].x, kv[j].y, kv[j].z, kv[j].w};
                unsigned cand = 0; // bit q = row q of this lane's eight
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned g = ((w[i] | 0x80008000u) - kp2) & (kn2 - w[i]) & 0x80008000u;
                    cand |= (((g >> 15) & 1u) | ((g >> 30) & 2u)) << (2 * i);
                }
                if (__ballot(cand != 0) == 0) continue;
                const long long r0 = t + j * kKeyRowsPerLoad + 8 * lane;
                unsigned hits = 0;
                for (unsigned m = cand; m; m &= m - 1) {
                    const int q = __ffs((int)m) - 1;
                    const unsigned ww = q < 4 ? (q < 2 ? kv[j].x : kv[j].y) : (q < 6 ? kv[j].z : kv[j].w); // no dynamic register indexing
                    const unsigned k = (ww >> ((q & 1) * 16)) & 0xFFFFu;
                    if (is_hit(k, r0 + q)) hits |= 1u << q;
                }
                // order-preserving placement: exclusive prefix of the per-lane hit counts
                const int c = __popc(hits);
                int incl = c;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int v = __shfl_up(incl, o, kWave);
                    if (lane >= o) incl += v;
                }
                int pos = fill + incl - c;
                for (unsigned m = hits; m; m &= m - 1) out[pos++] = (int)r0 + (__ffs((int)m) - 1);
                fill += __shfl(incl, kWave - 1, kWave);
            }
        } else {
            for (long long r0 = t; r0 < w1; r0 += kWave) {
                const long long r = r0 + lane;
                const bool h = r < w1 && is_hit(key[r], r);
                const unsigned long long b = __ballot(h);
                if (h) out[fill + prefix_in_ballot(b)] = (int)r;
                fill += __popcll(b);
            }
        }
    }
    return w0 < n ? fill : 0;
}

template <int UNROLL>
__global__ __launch_bounds__(kK1Threads) void k_expired_stage_keyed(const lkey_t* __restrict__ key, const long long* __restrict__ end,
                                                                    long long n, long long rows_per_block, long long prev_now,
                                                                    long long now, unsigned kp, unsigned kn,
                                                                    int* __restrict__ stage, int* __restrict__ wave_count)
{
    const int fill = expired_stage_keyed_body<UNROLL>(key, end, n, rows_per_block, prev_now, now, kp, kn, stage, (int)blockIdx.x);
    if ((threadIdx.x & (kWave - 1)) == 0) wave_count[blockIdx.x * kK1Waves + (threadIdx.x >> 6)] = fill;
}

__global__ __launch_bounds__(256) void k_expired_gather(const int* __restrict__ stage, const int* __restrict__ wave_count,
                                                        const long long* __restrict__ wave_off, int n_waves,
                                                        long long rows_per_wave, int* __restrict__ queue, long long cap)
{
    const int lane = threadIdx.x & 63;
    const int total_waves = gridDim.x * 4;
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n_waves; w += total_waves) {
        const int cnt = wave_count[w];
        const long long src = (long long)w * rows_per_wave, dst = wave_off[w];
        for (int i = lane; i < cnt; i += 64)
            if (dst + i < cap) queue[dst + i] = stage[src + i];
    }
}

// ------------------------------------------------------------------------------------------------ archive group-min chain

// Per group (key = user column): earliest = min(start) and first = min(row) over the rows present (not tombstoned):
// the `list.reduce(min)` and the Map insertion order of /root/reference/server/storage/sqlProvider.js:763-794.
// A row only issues an atomic when it improves the value it reads first (minima only decrease, so a stale read can
// only cause a redundant atomic, never a missed one): ~ln(group size) atomics per group instead of one per row.


// queue = concatenation of the qualifying groups' row lists in first-appearance order: group k (user grp_user[k]) owns
// queue[grp_off[k] .. grp_off[k+1]); its rows sit, already in row order, at idx[offsets[user] .. +counts[user])
// ---- the archive chain on the device (pie_archive_queue; /root/reference/server/storage/sqlProvider.js:758-816,834-861).
// No table pass gathers per-group state from L2 (a first device version kept min(start) and the first row of every group with
// guarded atomics: two L2 gathers per row, 1.6 ms for the statistics pass alone at cfg3):
//   * a group qualifies iff now - min(start) >= window  <=>  SOME live row of it has start <= now - window: the flag pass only
//     touches per-group state for rows that pass that compare (a bit in a bitmap, tested before it is set);
//   * the selection looks the bitmap up in LDS;
//   * first-appearance order needs no per-row work at all: after the STABLE sort of the selected (group, row) pairs by group the
//     head of every group's run is its first row.
// k_arch_flag: 16-byte loads, two rows per lane; `bits` is the qualifying-groups bitmap (n_users bits, zeroed by the caller)
__global__ __launch_bounds__(256) void k_arch_flag(const long long* __restrict__ start, const long long* __restrict__ end,
                                                   const int* __restrict__ user, long long n, int n_users, long long limit, bool none,
                                                   unsigned int* __restrict__ bits)
{
    if (none) return; // now - window below every int64: no start can qualify
    const long long pairs = n >> 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    auto mark = [&](long long r) {
        const int g = user[r]; // late: only rows that pass the compare need their group
        if ((unsigned)g >= (unsigned)n_users) return;
        const unsigned bit = 1u << (g & 31);
        if (!(bits[g >> 5] & bit)) atomicOr(&bits[g >> 5], bit); // the test reads through L1 (a stale line only repeats an idempotent OR)
    };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const ll2_t e = stream_load<true>(reinterpret_cast<const ll2_t*>(end) + i);
        const ll2_t sv = stream_load<true>(reinterpret_cast<const ll2_t*>(start) + i);
        if (e.x != INT64_MIN && sv.x <= limit) mark(2 * i);
        if (e.y != INT64_MIN && sv.y <= limit) mark(2 * i + 1);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0 && end[n - 1] != INT64_MIN && start[n - 1] <= limit) mark(n - 1);
}
__global__ __launch_bounds__(256) void k_arch_popc(const unsigned int* __restrict__ bits, int words, unsigned int* __restrict__ n_qual)
{
    __shared__ int s_w[4];
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    int c = i < words ? __popc(bits[i]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && s_w[0] + s_w[1] + s_w[2] + s_w[3]) atomicAdd(n_qual, (unsigned)(s_w[0] + s_w[1] + s_w[2] + s_w[3]));
}
// The selection, order-preserving, twice over (end, user) = 12 B/row: WRITE = false counts per block, true writes (group, row)
// pairs in table order behind the block's offset.  The bitmap sits in LDS (dynamic: `words` x 4 bytes) when it fits.
template <bool WRITE, bool LDS_BITS>
__global__ __launch_bounds__(256) void k_arch_select(const long long* __restrict__ end, const int* __restrict__ user, long long n, long long rows_per_block,
                                                     const unsigned int* __restrict__ bits, int words, int n_users, int* __restrict__ blk_count,
                                                     const long long* __restrict__ blk_off, unsigned int* __restrict__ keys, int* __restrict__ rows)
{
    extern __shared__ unsigned int l_bits[];
    __shared__ int wcount[4];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (LDS_BITS) {
        for (int i = threadIdx.x; i < words; i += 256) l_bits[i] = bits[i];
    }
    const long long c0 = (long long)blockIdx.x * rows_per_block; // rows_per_block is even
    const long long c1 = min(n, c0 + rows_per_block);
    if (threadIdx.x == 0) carry_s = WRITE ? blk_off[blockIdx.x] : 0;
    __syncthreads();
    const unsigned int* bt = LDS_BITS ? l_bits : bits;
    long long total = 0;
    for (long long r0 = c0; r0 < c1; r0 += 512) { // 256 threads x 2 rows
        const long long r = r0 + 2 * threadIdx.x;
        bool h0 = false, h1 = false;
        int g0 = 0, g1 = 0;
        if (r + 1 < c1) {
            const ll2_t e = stream_load<true>(reinterpret_cast<const ll2_t*>(end + r));
            const i2_t u = stream_load<true>(reinterpret_cast<const i2_t*>(user + r));
            g0 = u.x; g1 = u.y;
            h0 = e.x != INT64_MIN && (unsigned)g0 < (unsigned)n_users && ((bt[g0 >> 5] >> (g0 & 31)) & 1u);
            h1 = e.y != INT64_MIN && (unsigned)g1 < (unsigned)n_users && ((bt[g1 >> 5] >> (g1 & 31)) & 1u);
        } else if (r < c1) {
            g0 = user[r];
            h0 = end[r] != INT64_MIN && (unsigned)g0 < (unsigned)n_users && ((bt[g0 >> 5] >> (g0 & 31)) & 1u);
        }
        const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1);
        const int mine = (h0 ? 1 : 0) + (h1 ? 1 : 0);
        // rows of a lane are consecutive: position inside the wave = both rows of every lane below + this lane's own
        const int before = prefix_in_ballot(b0) + prefix_in_ballot(b1);
        const int wtotal = __popcll(b0) + __popcll(b1);
        if constexpr (WRITE) {
            if (lane == 0) wcount[wave] = wtotal;
            __syncthreads();
            long long base = carry_s;
            for (int w = 0; w < wave; ++w) base += wcount[w];
            long long pos = base + before;
            if (h0) { keys[pos] = (unsigned)g0; rows[pos] = (int)r; ++pos; }
            if (h1) { keys[pos] = (unsigned)g1; rows[pos] = (int)(r + 1); }
            __syncthreads();
            if (threadIdx.x == 0) carry_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        } else {
            (void)mine; (void)before;
            total += wtotal; // wave-uniform
        }
    }
    if constexpr (!WRITE) {
        if (lane == 0) wcount[wave] = (int)total;
        __syncthreads();
        if (threadIdx.x == 0) blk_count[blockIdx.x] = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    }
}
// after the stable sort by group: the head of a group's run is its first row, head and tail give its place and size
__global__ __launch_bounds__(256) void k_arch_heads(const unsigned int* __restrict__ g_sorted, const int* __restrict__ r_sorted, long long m,
                                                    int* __restrict__ ghead, int* __restrict__ gfirst, int* __restrict__ glast)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const unsigned g = g_sorted[i];
        if (i == 0 || g_sorted[i - 1] != g) { ghead[g] = (int)i; gfirst[g] = r_sorted[i]; }
        if (i == m - 1 || g_sorted[i + 1] != g) glast[g] = (int)i;
    }
}
// sort key of every group: its first row if it qualifies (its bit is set and it has a run), else behind everything
__global__ __launch_bounds__(256) void k_arch_group_keys(const unsigned int* __restrict__ bits, const int* __restrict__ gfirst, int n_users,
                                                         unsigned int* __restrict__ key, int* __restrict__ val)
{
    const int g = (int)(blockIdx.x * 256 + threadIdx.x);
    if (g >= n_users) return;
    const bool q = (bits[g >> 5] >> (g & 31)) & 1u;
    key[g] = q ? (unsigned)gfirst[g] : 0xFFFFFFFFu;
    val[g] = g;
}
// sizes of the qualifying groups in first-appearance order (entry k = group order[k])
__global__ __launch_bounds__(256) void k_arch_sizes(const int* __restrict__ order, const unsigned int* __restrict__ n_qual, const int* __restrict__ ghead,
                                                    const int* __restrict__ glast, unsigned int* __restrict__ size)
{
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k < *n_qual) size[k] = (unsigned)(glast[order[k]] - ghead[order[k]] + 1);
}
// group k's run (sorted rows at ghead[order[k]]) -> the queue at off[k]: one block per group, coalesced
__global__ __launch_bounds__(256) void k_arch_gather(const int* __restrict__ order, const unsigned int* __restrict__ n_qual, const int* __restrict__ ghead,
                                                     const unsigned int* __restrict__ size, const unsigned int* __restrict__ off,
                                                     const int* __restrict__ r_sorted, int* __restrict__ queue)
{
    const unsigned nq = *n_qual;
    for (unsigned k = blockIdx.x; k < nq; k += gridDim.x) {
        const unsigned src = (unsigned)ghead[order[k]], dst = off[k], cnt = size[k];
        for (unsigned i = threadIdx.x; i < cnt; i += 256) queue[dst + i] = r_sorted[src + i];
    }
}


} // namespace pie
