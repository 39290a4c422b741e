// pie_kernels.h — device kernels of the session-scan -> per-user feed path, written for gfx950 (CDNA4):
// 64-lane wavefronts, SoA columns streamed with 16-B / 8-B per-lane loads, per-wave ballot + mbcnt prefix
// compaction staged in LDS, one pass over the 24 B/row table.  Integer only — no MFMA on this path.
//
// Pipeline of one scan (all on one stream):
//   K1  k_scan_compact   read 4 columns once, predicate, histogram counts[user], compact selected rows
//   K2a k_tile_sums      per-tile sums of counts[U]  (+ prefix of K1's per-block record counts)
//   K2b k_offsets        exclusive scan -> offsets[U+1], work lists for the sort kernels, summary
//   K3  k_scatter        selected records -> per-user buckets (slot = offsets[u] + atomic rank)
//   K4a k_sort_tiny      buckets of <= TINY_MAX rows: one thread per bucket, rank sort
//   K4b k_sort_segments  buckets (or 4096-row tiles of big buckets): one block each, LDS bitonic
//   K4c k_merge_pass     big buckets only: log2(n/4096) rank-merge passes
// Order inside a bucket is (start asc, row index asc): ORDER BY start_ts ASC
// (/root/reference/server/storage/sqlProvider.js:276) with the tie rule of SURVEY.md §8 a-D.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pie {

struct alignas(16) SelRec {
    long long start;
    int idx;
    int user;
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
};

constexpr int kWave = 64;
constexpr int kK1Threads = 256;
constexpr int kK1Waves = kK1Threads / kWave;
constexpr int kUnitRows = 2 * kWave;            // one int64x2 load per lane
constexpr int kUnroll = 4;                      // units per wave-tile
constexpr int kWaveTileRows = kUnitRows * kUnroll; // 512 rows per wave per iteration
constexpr int kBlockTileRows = kWaveTileRows * kK1Waves;
constexpr int kStage = 128;                     // per-wave LDS ring of selected records
constexpr int kTinyMax = 16;
constexpr int kSegMax = 4096;
constexpr int kScanTile = 2048;                 // counts per K2 block (256 threads x 8)

// ------------------------------------------------------------------------------------------------ helpers

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `b` in lanes below the caller
__device__ __forceinline__ int prefix_in_ballot(unsigned long long b)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
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
        const long long s = T0 - (long long)__umul64hi(r2, (unsigned long long)SPAN);
        long long dur = TTL;
        if (flags & 1u) dur = MIN_DUR + (long long)__umul64hi(r3, (unsigned long long)(TTL - MIN_DUR + 1));
        start[k] = s;
        end[k] = s + dur;
        user[k] = (flags & 2u) ? (int)((i * (unsigned long long)n_users) / (unsigned long long)n_total)
                               : (int)__umul64hi(r0, (unsigned long long)n_users);
        disc[k] = (int)__umul64hi(r1, (unsigned long long)n_disc);
    }
}

// ------------------------------------------------------------------------------------------------ K1 scan + compact

struct WaveStage {
    SelRec* ring;  // this wave's LDS ring (kStage records)
    int head;      // wave-uniform
    int fill;      // wave-uniform, < 64 between calls
};

// Flush 64 staged records: one LDS atomic per wave reserves 64 slots of the block's private output range,
// every lane stores one 16-B record (1 KiB coalesced per wave).
__device__ __forceinline__ void stage_flush(WaveStage& st, int count, SelRec* __restrict__ out, int* blk_cursor, int lane)
{
    int base = 0;
    if (lane == 0) base = atomicAdd(blk_cursor, count);
    base = __builtin_amdgcn_readfirstlane(base);
    if (lane < count) out[base + lane] = st.ring[(st.head + lane) & (kStage - 1)];
    st.head = (st.head + count) & (kStage - 1);
    st.fill -= count;
}

__device__ __forceinline__ void consider_row(bool sel, long long s, int row, int u, WaveStage& st, int n_users,
                                             int* __restrict__ counts, SelRec* __restrict__ out, int* blk_cursor,
                                             unsigned int* bad_rows, int lane)
{
    const bool user_ok = (unsigned)u < (unsigned)n_users;
    if (sel & !user_ok) atomicAdd(bad_rows, 1u);
    sel &= user_ok;
    const unsigned long long b = __ballot(sel);
    if (b == 0) return; // wave-uniform: nothing selected in this 64-row slice
    if (sel) {
        atomicAdd(&counts[u], 1); // result unused -> no-return global_atomic_add
        SelRec r;
        r.start = s;
        r.idx = row;
        r.user = u;
        st.ring[(st.head + st.fill + prefix_in_ballot(b)) & (kStage - 1)] = r;
    }
    st.fill += __popcll(b);
    __builtin_amdgcn_wave_barrier();
    if (st.fill >= kWave) stage_flush(st, kWave, out, blk_cursor, lane);
    __builtin_amdgcn_wave_barrier();
}

// Block b owns rows [b*rows_per_block, (b+1)*rows_per_block) and the same index range of `sel` as its
// private output region (a block can never select more rows than it reads), so compaction needs no global
// cursor: blk_count[b] says how many records the region holds.
__global__ __launch_bounds__(kK1Threads) void k_scan_compact(
    const long long* __restrict__ start, const long long* __restrict__ end, const int* __restrict__ user,
    const int* __restrict__ disc, long long n, long long rows_per_block, long long now, long long cutoff,
    unsigned long long mask, int n_users, int* __restrict__ counts, SelRec* __restrict__ sel,
    int* __restrict__ blk_count, unsigned int* __restrict__ bad_rows)
{
    __shared__ SelRec stage[kK1Waves][kStage];
    __shared__ int blk_cursor;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) blk_cursor = 0;
    __syncthreads();

    const long long c0 = (long long)blockIdx.x * rows_per_block;
    long long c1 = c0 + rows_per_block;
    if (c1 > n) c1 = n;
    SelRec* out = sel + c0;
    WaveStage st;
    st.ring = stage[wave];
    st.head = 0;
    st.fill = 0;

    for (long long t = c0 + (long long)wave * kWaveTileRows; t < c1; t += kBlockTileRows) {
        if (t + kWaveTileRows <= c1) {
            // full wave-tile: 4 units x (2 rows per lane); every load is a fully used, aligned line
            longlong2 s[kUnroll], e[kUnroll];
            int2 u[kUnroll], d[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const long long r = t + j * kUnitRows + 2 * lane;
                e[j] = *reinterpret_cast<const longlong2*>(end + r);
                s[j] = *reinterpret_cast<const longlong2*>(start + r);
                d[j] = *reinterpret_cast<const int2*>(disc + r);
                u[j] = *reinterpret_cast<const int2*>(user + r);
            }
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int r = (int)(t + j * kUnitRows + 2 * lane);
                consider_row(row_selected(s[j].x, e[j].x, d[j].x, now, cutoff, mask), s[j].x, r, u[j].x, st, n_users,
                             counts, out, &blk_cursor, bad_rows, lane);
                consider_row(row_selected(s[j].y, e[j].y, d[j].y, now, cutoff, mask), s[j].y, r + 1, u[j].y, st,
                             n_users, counts, out, &blk_cursor, bad_rows, lane);
            }
        } else {
            // ragged tail of the block's range: one row per lane, bounds-checked
            const long long t1 = (t + kWaveTileRows < c1) ? t + kWaveTileRows : c1;
            for (long long r0 = t; r0 < t1; r0 += kWave) {
                const long long r = r0 + lane;
                const bool in = r < t1;
                long long sv = 0, ev = 0;
                int uv = 0, dv = -1;
                if (in) {
                    sv = start[r];
                    ev = end[r];
                    uv = user[r];
                    dv = disc[r];
                }
                consider_row(in && row_selected(sv, ev, dv, now, cutoff, mask), sv, (int)r, uv, st, n_users, counts, out,
                             &blk_cursor, bad_rows, lane);
            }
        }
    }
    if (st.fill > 0) stage_flush(st, st.fill, out, &blk_cursor, lane);
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = blk_cursor;
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

// K2a: tile_sum[b] = sum of counts over tile b.  The extra last block turns K1's per-block record counts
// into the prefix blk_off[nb+1] that K3 uses to balance its reads.
__global__ __launch_bounds__(256) void k_tile_sums(const int* __restrict__ counts, int n_users,
                                                   long long* __restrict__ tile_sum, int n_tiles,
                                                   const int* __restrict__ blk_count, int nb,
                                                   long long* __restrict__ blk_off, Summary* __restrict__ summary)
{
    __shared__ long long lds4[4];
    if ((int)blockIdx.x < n_tiles) {
        const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
        long long v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k < n_users) v += counts[base + k];
        v = block_sum_256(v, lds4);
        if (threadIdx.x == 0) tile_sum[blockIdx.x] = v;
        return;
    }
    // last block: exclusive prefix over blk_count[0..nb)
    __shared__ long long wsum[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const long long c = b < nb ? blk_count[b] : 0;
        const long long incl = wave_incl_scan(c, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        long long wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const long long carry = carry_s;
        if (b < nb) blk_off[b] = carry + wbase + incl - c;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blk_off[nb] = carry_s;
        summary->m = (unsigned long long)carry_s;
    }
}

// K2b: offsets[u] = exclusive prefix of counts; cursor[u] = 0; sort work lists; max bucket.
__global__ __launch_bounds__(256) void k_offsets(const int* __restrict__ counts, int n_users,
                                                 const long long* __restrict__ tile_sum,
                                                 long long* __restrict__ offsets, int* __restrict__ cursor,
                                                 Segment* __restrict__ seg_list, int* __restrict__ big_list,
                                                 Summary* __restrict__ summary)
{
    __shared__ long long lds4[4];
    __shared__ long long wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // base = sum of the tile sums in front of this tile (n_tiles is small: U / 2048)
    long long part = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_sum[t];
    const long long base = block_sum_256(part, lds4);

    const int u0 = blockIdx.x * kScanTile + threadIdx.x * 8;
    int c[8];
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c[k] = (u0 + k < n_users) ? counts[u0 + k] : 0;
        tsum += c[k];
    }
    const long long incl = wave_incl_scan(tsum, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long run = base + incl - tsum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    unsigned int local_max = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int u = u0 + k;
        if (u < n_users) {
            offsets[u] = run;
            cursor[u] = 0;
            const int n = c[k];
            local_max = max(local_max, (unsigned)n);
            if (n > kTinyMax) {
                if (n <= kSegMax) {
                    const unsigned slot = atomicAdd(&summary->n_seg, 1u);
                    Segment sg;
                    sg.pos = run;
                    sg.len = n;
                    sg.flags = 0;
                    seg_list[slot] = sg;
                } else {
                    const int tiles = (n + kSegMax - 1) / kSegMax;
                    const unsigned slot = atomicAdd(&summary->n_seg, (unsigned)tiles);
                    for (int t = 0; t < tiles; ++t) {
                        Segment sg;
                        sg.pos = run + (long long)t * kSegMax;
                        sg.len = min(kSegMax, n - t * kSegMax);
                        sg.flags = 1;
                        seg_list[slot + t] = sg;
                    }
                    big_list[atomicAdd(&summary->n_big, 1u)] = u;
                }
            }
            run += n;
        }
    }
    if (u0 + 8 >= n_users && u0 < n_users) offsets[n_users] = run; // thread holding the last user
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_max = max(local_max, (unsigned)__shfl_xor((int)local_max, o, kWave));
    if (lane == 0 && local_max > 0) atomicMax(&summary->max_count, local_max);
}

// ------------------------------------------------------------------------------------------------ K3 scatter

// Record r of the concatenated per-block regions -> bucket slot offsets[user] + rank, rank from a returning
// atomic on cursor[user].  Slot order inside a bucket is arbitrary here; K4 makes it (start, idx) order.
__global__ __launch_bounds__(256) void k_scatter(const SelRec* __restrict__ sel, const long long* __restrict__ blk_off,
                                                 int nb, long long rows_per_block,
                                                 const long long* __restrict__ offsets, int* __restrict__ cursor,
                                                 long long* __restrict__ bkt_start, int* __restrict__ bkt_idx)
{
    const long long m = blk_off[nb];
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nb; // last b with blk_off[b] <= r
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (blk_off[mid] <= r) lo = mid; else hi = mid;
        }
        const SelRec rec = sel[(long long)lo * rows_per_block + (r - blk_off[lo])];
        const int rank = atomicAdd(&cursor[rec.user], 1);
        const long long pos = offsets[rec.user] + rank;
        bkt_start[pos] = rec.start;
        bkt_idx[pos] = rec.idx;
    }
}

// ------------------------------------------------------------------------------------------------ K4 order

// K4a: one thread per bucket of <= kTinyMax rows: rank sort (keys (start, idx) are unique).
__global__ __launch_bounds__(256) void k_sort_tiny(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                   int n_users, const long long* __restrict__ bkt_start,
                                                   const int* __restrict__ bkt_idx, int* __restrict__ out_idx)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_users) return;
    const int n = counts[u];
    if (n == 0 || n > kTinyMax) return;
    const long long o = offsets[u];
    for (int j = 0; j < n; ++j) {
        const long long sj = bkt_start[o + j];
        const int ij = bkt_idx[o + j];
        int rank = 0;
        for (int k = 0; k < n; ++k) rank += key_less(bkt_start[o + k], bkt_idx[o + k], sj, ij) ? 1 : 0;
        out_idx[o + rank] = ij;
    }
}

// K4b: one block per segment (<= kSegMax rows): bitonic sort of (start, idx) in LDS.
__global__ __launch_bounds__(256) void k_sort_segments(const Segment* __restrict__ seg_list,
                                                       const Summary* __restrict__ summary,
                                                       long long* __restrict__ bkt_start, int* __restrict__ bkt_idx,
                                                       int* __restrict__ out_idx)
{
    __shared__ long long ks[kSegMax];
    __shared__ int ki[kSegMax];
    const unsigned n_seg = summary->n_seg;
    for (unsigned w = blockIdx.x; w < n_seg; w += gridDim.x) {
        const Segment sg = seg_list[w];
        int p = 32;
        while (p < sg.len) p <<= 1;
        for (int i = threadIdx.x; i < p; i += blockDim.x) {
            const bool in = i < sg.len;
            ks[i] = in ? bkt_start[sg.pos + i] : INT64_MAX;
            ki[i] = in ? bkt_idx[sg.pos + i] : INT32_MAX;
        }
        __syncthreads();
        for (int k = 2; k <= p; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (p >> 1); t += blockDim.x) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // index with bit j clear
                    const int l = i | j;
                    const bool up = (i & k) == 0;
                    const long long sa = ks[i], sb = ks[l];
                    const int ia = ki[i], ib = ki[l];
                    const bool swap = up ? key_less(sb, ib, sa, ia) : key_less(sa, ia, sb, ib);
                    if (swap) {
                        ks[i] = sb; ks[l] = sa;
                        ki[i] = ib; ki[l] = ia;
                    }
                }
                __syncthreads();
            }
        }
        if (sg.flags & 1) {
            for (int i = threadIdx.x; i < sg.len; i += blockDim.x) {
                bkt_start[sg.pos + i] = ks[i];
                bkt_idx[sg.pos + i] = ki[i];
            }
        } else {
            for (int i = threadIdx.x; i < sg.len; i += blockDim.x) out_idx[sg.pos + i] = ki[i];
        }
        __syncthreads();
    }
}

// K4c: one merge pass over every big bucket: sorted runs of `width` -> sorted runs of 2*width.  Each thread
// owns one input element, binary-searches its rank in the sibling run, and stores it at its final slot
// (keys are unique, so ranks are a permutation).  grid.y indexes big_list.
__global__ __launch_bounds__(256) void k_merge_pass(const int* __restrict__ big_list, int n_big, const int* __restrict__ counts,
                                                    const long long* __restrict__ offsets, long long width,
                                                    const long long* __restrict__ src_s, const int* __restrict__ src_i,
                                                    long long* __restrict__ dst_s, int* __restrict__ dst_i)
{
    for (int bb = blockIdx.y; bb < n_big; bb += gridDim.y) {
    const int u = big_list[bb];
    const long long n = counts[u], o = offsets[u];
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const long long run = p / width;
        const long long pair0 = (run >> 1) * 2 * width; // first slot of this pair of runs
        const bool right = run & 1;
        const long long sib0 = right ? pair0 : pair0 + width;
        long long sib_n = n - sib0;
        if (sib_n > width) sib_n = width;
        if (sib_n < 0) sib_n = 0;
        const long long s = src_s[o + p];
        const int i = src_i[o + p];
        long long lo = 0, hi = sib_n; // number of sibling keys smaller than (s, i)
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (key_less(src_s[o + sib0 + mid], src_i[o + sib0 + mid], s, i)) lo = mid + 1; else hi = mid;
        }
        const long long in_run = p - run * width;
        const long long q = o + pair0 + in_run + lo;
        dst_s[q] = s;
        dst_i[q] = i;
    }
    }
}

// ------------------------------------------------------------------------------------------------ table maintenance

__global__ __launch_bounds__(256) void k_set_end(long long* __restrict__ end, const int* __restrict__ rows,
                                                 const long long* __restrict__ new_end, long long k, long long n)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k && (unsigned)rows[t] < (unsigned long long)n) end[rows[t]] = new_end[t];
}

// deleteSessionsForUser (/root/reference/server/sessionStore.js:55-64): strict user match, tombstone = never live
__global__ __launch_bounds__(256) void k_delete_user(const int* __restrict__ user, long long* __restrict__ end, long long n,
                                                     int target, unsigned long long* __restrict__ n_deleted)
{
    unsigned long long local = 0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        if (user[r] == target && end[r] != INT64_MIN) {
            end[r] = INT64_MIN;
            ++local;
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, kWave);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_deleted, local);
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

// ------------------------------------------------------------------------------------------------ expired queue ("next" row)

// change predicate prev_now < end <= now, order-preserving compaction in three steps:
// per-block counts -> prefix (single block) -> ordered write.
__global__ __launch_bounds__(256) void k_expired_count(const long long* __restrict__ end, long long n, long long rows_per_block,
                                                       long long prev_now, long long now, int* __restrict__ blk_count)
{
    __shared__ long long lds4[4];
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    const long long c1 = min(n, c0 + rows_per_block);
    long long local = 0;
    for (long long r = c0 + threadIdx.x; r < c1; r += blockDim.x) {
        const long long e = end[r];
        local += (e <= now && e > prev_now) ? 1 : 0;
    }
    local = block_sum_256(local, lds4);
    if (threadIdx.x == 0) blk_count[blockIdx.x] = (int)local;
}

__global__ __launch_bounds__(256) void k_expired_write(const long long* __restrict__ end, long long n, long long rows_per_block,
                                                       long long prev_now, long long now, const long long* __restrict__ blk_off,
                                                       int* __restrict__ queue, long long cap)
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
        bool hit = false;
        if (r < c1) {
            const long long e = end[r];
            hit = (e <= now && e > prev_now);
        }
        const unsigned long long b = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(b);
        __syncthreads();
        long long base = carry_s;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        const long long pos = base + prefix_in_ballot(b);
        if (hit && pos < cap) queue[pos] = (int)r;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
}

} // namespace pie
