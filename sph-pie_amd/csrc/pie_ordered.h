// The ordered run: the table's rows a second time, in the order every answer wants them.
//
// A feed scan returns, per user, the selected rows ordered by (start, row).  The general path finds that grouping and
// order anew for every query: one returning histogram atomic per selected row (25 G/s chip-wide, whatever the scope:
// DESIGN.md "What the atomics cost") and a per-bucket sort — fine for the sparse spec query, 1.8 ms when a quarter of a
// 10^8-row table is selected, and at the mercy of the user distribution (a Zipf head user is one bucket of 41 k rows).
// None of that work depends on the query: (user, start, row) is a property of the TABLE.  The ordered run stores the
// rows once in that order — the result of the all-selecting scan, i.e. built by the general path itself — as
//     o_pay[i] = {start, row, disc}   o_end[i]   o_key[i] / o_fkey[i] (the liveness keys of o_end)   pos[row] = i
// plus uoff[u], the first position of user u's segment (its rows, then spare slots holding filler records: key 0,
// discipline -1, never selected).  A query is then a FILTER over positions 0 .. n_ord: the selected positions,
// in position order, ARE the answer (idx = their row ids, offsets[u] = selected positions before uoff[u]).  No atomics,
// no sort, no dependence on how the rows are spread over the users; every kernel below is a straight pass:
//
//   k_ord_scan_dense    every position's 2-byte key; the 16-byte record of every 64-position slice that holds a row that may be
//                       live (`end` only for ambiguous keys): at most 18 B/row, less the more rows are dead; per 4096-position
//                       tile the selected row ids go to the staging array in order, with the tile's count and its 64 slice
//                       ballots / prefixes (what a rank lookup needs)
//   k_ord_scan_keyed    the sparse form: streams only the key column (1 or 2 B/row), candidates of a chunk queue up IN POSITION
//                       ORDER in a per-wave LDS ring and are evaluated 64 at a time (one 16-byte gather each); the selected
//                       positions of a chunk go to the staging array in order, with the chunk's count.  Chunks are dealt to
//                       the waves round robin (a dense stretch — the head user's live rows — spreads over as many waves as it
//                       has chunks)
//   k_ord_prefix        exclusive prefix of the unit (tile / chunk) counts, two levels in one launch
//   k_ord_emit          the copy staging -> idx (balanced per element / per tile) and, in other blocks of the same launch,
//                       offsets[u] = rank of uoff[u] among the selected positions, counts, largest bucket
//   k_ord_publish       one wave: summary to the host
//
// No kernel waits for another block: there is no look-back, no ticket, no spin anywhere in this file ("the block that
// finishes last does X" is a counter, not a wait).
//
// Writers of `end` (touch, delete, purge) mirror their store into o_end / the keys through pos[] (OrdMirror in
// pie_kernels.h).  Appends are inserted at their place in their user's segment, which ends in spare slots (k_ord_append: a
// session store's rows arrive in time order, i.e. at the end; a late row shifts the few behind it).  A load, a re-shard or a
// row that arrives far out of order (a back-fill) invalidates the run and the host falls back to the general path until it
// is rebuilt; a full segment is handled by a re-spread.
#pragma once

namespace pie {

struct alignas(16) OrdRec {
    long long start;
    int row;
    int disc;
};

constexpr int kOrdTile = 4096;       // positions per unit of the dense form
constexpr int kOrdTileShift = 12;
constexpr int kOrdSlices = kOrdTile / kWave;

struct OrdCtl {
    unsigned int done_prefix; // blocks of the prefix kernel that have finished
    unsigned int pad[3];
};

// spare slots of the run carry end = PIE_END_NONE (never live, key 0 under every key base — a memset 0 would read as a live row
// once the table's `end` values are all <= 0: ADVICE r02)
__global__ __launch_bounds__(256) void k_fill_ll(long long* __restrict__ p, long long n, long long v)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// out_idx of the all-selecting scan -> the run's columns.  Entry i of the sorted list belongs to user u = pay[row].user and is
// its (i - off[u])-th row; it goes to position uoff[u] + that rank: every user's segment starts at uoff[u] and ends in spare
// slots (see k_ord_append).
__global__ __launch_bounds__(256) void k_ord_gather(const int* __restrict__ idx, long long m, const long long* __restrict__ off,
                                                    const long long* __restrict__ uoff, const PayRec* __restrict__ pay,
                                                    const long long* __restrict__ end, long long key_base, int key_shift,
                                                    long long fkey_base, int fkey_shift, OrdRec* __restrict__ o_pay,
                                                    long long* __restrict__ o_end, lkey_t* __restrict__ o_key,
                                                    fkey_t* __restrict__ o_fkey, int* __restrict__ pos)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const int r = idx[i];
        const PayRec p = pay[r];
        const long long ev = end[r];
        const long long at = uoff[p.user] + (i - off[p.user]);
        OrdRec o;
        o.start = p.start;
        o.row = r;
        o.disc = p.disc;
        o_pay[at] = o;
        o_end[at] = ev;
        // the keys are a function of `end` and the table's key parameters: computed, not gathered (each gather of a scattered
        // row costs a 128-byte fetch whatever it reads: two of them instead of four)
        o_key[at] = (lkey_t)key_of(ev, key_base, key_shift);
        o_fkey[at] = (fkey_t)key_of(ev, fkey_base, fkey_shift, kFineKeyMax);
        pos[r] = (int)at;
    }
}

// createSession on a table that has a run.  A user's segment ends in spare slots (a sixteenth of its rows + 16, filler records:
// key 0, discipline -1, never selected).  A new row of user u is INSERTED into u's segment at its place in (start, row)
// order — behind every row whose start is not above its own: the new row has the largest row id.  A session store creates
// sessions now, so that place is almost always the end of the segment (one compare); a row that arrives a little late
// (clock jitter between front-ends, a batch that is not sorted, a corpus whose last sessions lie "after now") shifts the
// few rows behind it by one slot.  All rows of one user in a batch are handled by ONE thread, so rows of the same user never
// race; different users are different threads.  k_ord_append_mark threads the batch's rows of each user on a chain (one
// atomicExch per row: bhead[u] = the row that came last, bnext[row] = the one before it); the thread of the chain's head walks
// it.  The chain's order is whatever order the exchanges took, so a row's place is found on (start, row), not on start alone:
// the rows of one user then end up in order however the chain runs.
// O(batch) + the shifts per row of the batch, nothing per row of the table — and nothing per row of the batch either: a thread
// that searched the batch for its user's later rows made the kernel as slow as its longest search (80 us per 1 000 rows).
//   placed[t] = pass   the row went into the run in this pass
//   stale[0]++         the row would have to move more than kOrdShiftMax rows (a back-fill, not a session store's append):
//                      the host drops the run
//   stale[1]++, pend[u]++   its user's segment is full: the host re-spreads (k_ord_respread gives every segment fresh spare
//                      slots), then pass 2 places the rows left over
// stale[] is mapped host memory, read by the host after the append's one synchronisation.
constexpr int kOrdShiftMax = 256;

// first pass of an append: the chains (bhead[] rests at -1 between appends: the thread that walks a chain puts it back)
__global__ __launch_bounds__(256) void k_ord_append_mark(const int* __restrict__ st_user, int k, int n_users, const int* __restrict__ placed_in,
                                                         int pass, int* __restrict__ bhead, int* __restrict__ bnext)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= k) return;
    const int u = st_user[t];
    if ((unsigned)u >= (unsigned)n_users || (pass > 1 && placed_in[t] != 0)) return;
    bnext[t] = atomicExch(&bhead[u], t);
}

__global__ __launch_bounds__(256) void k_ord_append(const long long* __restrict__ st_start, const long long* __restrict__ st_end,
                                                    const int* __restrict__ st_user, const int* __restrict__ st_disc, int k,
                                                    long long row0, int n_users, long long key_base, int key_shift, long long fkey_base,
                                                    int fkey_shift, const long long* __restrict__ uoff, int* __restrict__ ufill,
                                                    OrdRec* __restrict__ o_pay, long long* __restrict__ o_end, lkey_t* __restrict__ o_key,
                                                    fkey_t* __restrict__ o_fkey, int* __restrict__ pos, unsigned int* __restrict__ stale,
                                                    const int* __restrict__ placed_in, int* __restrict__ placed, int* __restrict__ pend, int pass,
                                                    int* __restrict__ bhead, const int* __restrict__ bnext)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= k) return;
    const int u = st_user[t];
    if ((unsigned)u >= (unsigned)n_users) { // pass 1: the whole append is rejected by the caller
        if (pass == 1) placed[t] = 0;
        return;
    }
    if (pass > 1 && placed_in[t] != 0) return; // pass 2: only what pass 1 left over
    if (bhead[u] != t) return;                 // another row of the batch heads my user's chain: its thread places mine too
    bhead[u] = -1;                             // back to rest for the next append (only this thread touches user u now)
    int fill = ufill[u];
    const long long seg = uoff[u];
    const long long cap = uoff[u + 1] - seg;
    auto place = [&](int e) { // insert row e of the batch into my user's segment
        const long long sv = st_start[e];
        const int row = (int)(row0 + e);
        if (fill >= cap) {
            placed[e] = 0;
            if (pend) pend[u] += 1;
            __hip_atomic_fetch_add(&stale[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        int at = fill; // behind every row that is before it in (start, row): usually the end of the segment
        if (fill > 0) {
            const OrdRec last = o_pay[seg + fill - 1];
            if (!key_less(last.start, last.row, sv, row)) {
                int lo = 0, hi = fill - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const OrdRec m = o_pay[seg + mid];
                    if (key_less(m.start, m.row, sv, row)) lo = mid + 1;
                    else hi = mid;
                }
                at = lo;
            }
        }
        if (fill - at > kOrdShiftMax) {
            placed[e] = 0;
            __hip_atomic_fetch_add(&stale[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        for (int i = fill - 1; i >= at; --i) { // make room: the rows behind the new one move up by one slot
            const OrdRec m = o_pay[seg + i];
            o_pay[seg + i + 1] = m;
            o_end[seg + i + 1] = o_end[seg + i];
            o_key[seg + i + 1] = o_key[seg + i];
            o_fkey[seg + i + 1] = o_fkey[seg + i];
            pos[m.row] = (int)(seg + i + 1);
        }
        const long long ev = st_end[e];
        OrdRec o;
        o.start = sv;
        o.row = row;
        o.disc = st_disc[e];
        o_pay[seg + at] = o;
        o_end[seg + at] = ev;
        o_key[seg + at] = (lkey_t)key_of(ev, key_base, key_shift);
        o_fkey[seg + at] = (fkey_t)key_of(ev, fkey_base, fkey_shift, kFineKeyMax);
        pos[row] = (int)(seg + at);
        placed[e] = pass;
        ++fill;
    };
    // the chain in batch order (then a row's place is the end of the segment and nothing shifts).  One row, nearly always; up to
    // four are sorted in registers; a longer chain (one user, five sessions in one batch) is read off the batch itself.
    int c0 = t, c1 = bnext[t], c2 = -1, c3 = -1, more = -1;
    if (c1 >= 0) {
        c2 = bnext[c1];
        if (c2 >= 0) {
            c3 = bnext[c2];
            if (c3 >= 0) more = bnext[c3];
        }
    }
    if (c1 < 0) place(c0);
    else if (more < 0) {
        constexpr int kNone = 0x7FFFFFFF;
        if (c2 < 0) c2 = kNone;
        if (c3 < 0) c3 = kNone;
        auto order = [](int& a, int& b) {
            const int lo = min(a, b), hi = max(a, b);
            a = lo;
            b = hi;
        };
        order(c0, c1);
        order(c2, c3);
        order(c0, c2);
        order(c1, c3);
        order(c1, c2);
        place(c0);
        place(c1);
        if (c2 != kNone) place(c2);
        if (c3 != kNone) place(c3);
    } else {
        for (int b = 0; b < k; b += 64) { // 64 users of the batch per step (the reads past k stay inside the staging block)
            const int4* v = reinterpret_cast<const int4*>(st_user + b);
            unsigned long long hit = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int4 x = v[j];
                hit |= (unsigned long long)((x.x == u) | ((x.y == u) << 1) | ((x.z == u) << 2) | ((x.w == u) << 3)) << (4 * j);
            }
            while (hit) {
                const int e = b + __builtin_ctzll(hit);
                hit &= hit - 1;
                if (e < k && (pass == 1 || placed_in[e] == 0)) place(e);
            }
        }
    }
    ufill[u] = fill;
}

// Fresh spare slots for every segment: the rows of the run move from (uoff_old, *_old) to (uoff_new, *_new), each user's rows
// staying in order at the head of its new segment.  A linear pass over the positions — no sort, nothing of the table is read.
__global__ __launch_bounds__(256) void k_ord_respread(long long n_old, int seg_users, const long long* __restrict__ uoff_old,
                                                      const int* __restrict__ ufill, const long long* __restrict__ uoff_new,
                                                      const OrdRec* __restrict__ pay_old, const long long* __restrict__ end_old,
                                                      const lkey_t* __restrict__ key_old, const fkey_t* __restrict__ fkey_old,
                                                      OrdRec* __restrict__ pay_new, long long* __restrict__ end_new,
                                                      lkey_t* __restrict__ key_new, fkey_t* __restrict__ fkey_new, int* __restrict__ pos)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n_old; p += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = seg_users - 1; // the user whose segment holds p: the largest u with uoff_old[u] <= p
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (uoff_old[mid] <= p) lo = mid;
            else hi = mid - 1;
        }
        const long long j = p - uoff_old[lo];
        if (j >= ufill[lo]) continue; // a spare slot
        const long long at = uoff_new[lo] + j;
        const OrdRec o = pay_old[p];
        pay_new[at] = o;
        end_new[at] = end_old[p];
        key_new[at] = key_old[p];
        fkey_new[at] = fkey_old[p];
        pos[o.row] = (int)at;
    }
}

// ------------------------------------------------------------------------------------------------ dense form

__global__ __launch_bounds__(256) void k_ord_scan_dense(const OrdRec* __restrict__ pay, const long long* __restrict__ end,
                                                        const lkey_t* __restrict__ key, long long n_ord, long long now,
                                                        unsigned now_key, long long cutoff, unsigned long long mask,
                                                        unsigned int* __restrict__ stage, int* __restrict__ unit_count,
                                                        unsigned long long* __restrict__ tile_ballot,
                                                        unsigned int* __restrict__ tile_prefix, Summary* __restrict__ summary)
{
    __shared__ int lrow[kOrdTile];
    __shared__ unsigned long long sball[kOrdSlices];
    __shared__ int wcnt[4];
    __shared__ int blk_live, blk_amb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_live = 0; blk_amb = 0; }
    int nlive = 0, namb = 0;
    int* mine = lrow + wave * 1024;
    const long long n_tiles = (n_ord + kOrdTile - 1) >> kOrdTileShift;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long w0 = (t << kOrdTileShift) + (long long)wave * 1024;
        int running = 0;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            // keys first (2 B per position), records only for the 64-position slices that hold a row that may be live: in run
            // order a user's rows ascend in start — and with it in end — so the rows a query finds dead are long stretches
            // at the head of every segment, and their 16-byte records are never fetched
            unsigned k[8];
            unsigned long long may[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long p = w0 + (h * 8 + j) * 64 + lane;
                k[j] = p < n_ord ? (unsigned)__builtin_nontemporal_load(key + p) : 0u;
            }
            OrdRec r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long p = w0 + (h * 8 + j) * 64 + lane;
                may[j] = __ballot(p < n_ord && k[j] >= now_key);
                r[j].start = 0;
                r[j].row = 0;
                r[j].disc = -1;
                if (may[j] != 0 && p < n_ord) { // wave-uniform test first: a dead slice costs no load at all
                    const ll2_t raw = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(pay + p));
                    r[j].start = raw.x;
                    r[j].row = (int)(raw.y & 0xFFFFFFFFll);
                    r[j].disc = (int)(raw.y >> 32);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long p = w0 + (h * 8 + j) * 64 + lane;
                if (may[j] == 0) { // nothing of this slice can be selected
                    if (lane == 0) sball[wave * 16 + h * 8 + j] = 0ull;
                    continue;
                }
                const bool valid = p < n_ord;
                const bool amb = valid && k[j] == now_key;
                bool live = valid && k[j] > now_key;
                if (amb) live = end[p] > now;
                const int dv = r[j].disc;
                const bool sel = live && r[j].start >= cutoff && (unsigned)dv < 64u && (((mask >> (dv & 63)) & 1ull) != 0);
                const unsigned long long b = __ballot(sel);
                if (sel) mine[running + prefix_in_ballot(b)] = r[j].row;
                running += __popcll(b);
                if (lane == 0) sball[wave * 16 + h * 8 + j] = b;
                nlive += __popcll(__ballot(live));
                namb += __popcll(__ballot(amb));
            }
        }
        if (lane == 0) wcnt[wave] = running;
        __syncthreads();
        int base_w = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int cw = wcnt[w];
            if (w < wave) base_w += cw;
            total += cw;
        }
        unsigned int* dst = stage + (t << kOrdTileShift) + base_w;
        for (int j = lane; j < running; j += 64) dst[j] = (unsigned)mine[j];
        if (wave == 0) {
            const unsigned long long b = sball[lane];
            const int cnt = __popcll(b);
            const int incl = wave_incl_scan_i32(cnt, lane);
            tile_ballot[t * kOrdSlices + lane] = b;
            tile_prefix[t * kOrdSlices + lane] = (unsigned)(incl - cnt);
            if (lane == 0) unit_count[t] = total;
        }
        __syncthreads();
    }
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    if (lane == 0 && namb) atomicAdd(&blk_amb, namb);
    __syncthreads();
    if (threadIdx.x == 0) add_row_stats(summary, blk_live, blk_amb);
}

// ------------------------------------------------------------------------------------------------ keyed (sparse) form

// A chunk is 512 positions for both key widths (8 keys per lane: one 8-byte or one 16-byte load).  The key arrays are padded
// with zero keys to a whole number of chunks; a padding position can only become a candidate when key(now) == 0, and is dropped
// when its batch is formed (pos >= n_ord).
template <class KT, int UNROLL = 4>
__global__ __launch_bounds__(256) void k_ord_scan_keyed(const OrdRec* __restrict__ pay, const long long* __restrict__ end,
                                                        const KT* __restrict__ key, long long n_ord, long long n_chunks,
                                                        long long now, unsigned now_key, long long cutoff, unsigned long long mask,
                                                        unsigned int* __restrict__ stage, int* __restrict__ unit_count,
                                                        Summary* __restrict__ summary)
{
    constexpr int kPerLane = 8; // 512 positions per chunk for both key widths: the candidate ring stays at 4 KiB per wave
    constexpr int kChunk = kPerLane * kWave;
    constexpr int kChunkShift = 9;
    constexpr int kRing = 2 * kChunk;
    constexpr int kUnroll = UNROLL;
    __shared__ int ring_s[4][kRing];
    __shared__ int blk_live, blk_amb, blk_cand, blk_chunk_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_live = 0; blk_amb = 0; blk_cand = 0; blk_chunk_max = 0; }
    __syncthreads();
    int* ring = ring_s[wave];
    int head = 0, fill = 0;                    // wave-uniform
    int nlive = 0, namb = 0, ncand = 0, chunk_max = 0;
    int cur_chunk = -1, cur_cnt = 0;           // the chunk of the last selected position so far and its count (not yet written)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);

    bool a_have = false, a_valid = false, a_amb = false;
    int a_pos = 0x7FFFFFFF;
    OrdRec a_pay;
    a_pay.start = 0; a_pay.row = 0; a_pay.disc = -1;
    long long a_end = 0;

    auto step_b = [&]() { // the gathers of the batch formed one step ago are back: evaluate, emit in position order
        if (!a_have) return;
        a_have = false;
        bool live = false, sel = false;
        if (a_valid) {
            live = a_amb ? (a_end > now) : true;
            const int dv = a_pay.disc;
            sel = live && a_pay.start >= cutoff && (unsigned)dv < 64u && (((mask >> (dv & 63)) & 1ull) != 0);
        }
        nlive += __popcll(__ballot(live));
        namb += __popcll(__ballot(a_valid && a_amb));
        ncand += __popcll(__ballot(a_valid));
        const unsigned long long sb = __ballot(sel);
        if (sb == 0) return;
        // positions ascend with the lane, so the lanes of one chunk are consecutive: a segment starts where the chunk id changes
        const int ch = a_pos >> kChunkShift;
        const int ch_below = __shfl_up(ch, 1, kWave);
        const unsigned long long hb = __ballot(lane == 0 || ch != ch_below);
        const int first = 63 - __clzll((long long)(hb & le));                       // first lane of my segment
        int rank = __popcll(sb & lt & ~((1ull << first) - 1ull));
        if (ch == cur_chunk) rank += cur_cnt;
        const unsigned long long ha = hb & ~le;                                     // segment heads above me
        const int seg_end = ha ? __ffsll((long long)ha) - 1 : 64;
        const unsigned long long upto = seg_end == 64 ? ~0ull : ((1ull << seg_end) - 1ull);
        const bool last_of_seg = sel && (sb & ~le & upto) == 0;
        const int last = 63 - __clzll((long long)sb);                               // last selected lane of the batch
        const int new_chunk = __shfl(ch, last, kWave);
        const int new_cnt = __shfl(rank, last, kWave) + 1;
        if (sel) {
            stage[((long long)ch << kChunkShift) + rank] = (unsigned)a_pos;
            // a chunk's count is written exactly once: here when no later batch can hold more of it, else it is carried
            if (last_of_seg && ch != new_chunk) unit_count[ch] = rank + 1;
        }
        // the carried chunk got nothing in this batch and the batch moved past it: its count is final
        if (cur_chunk >= 0 && cur_chunk != new_chunk && __ballot(sel && ch == cur_chunk) == 0 && lane == 0) unit_count[cur_chunk] = cur_cnt;
        cur_chunk = new_chunk;
        cur_cnt = new_cnt;
    };
    auto step_a = [&](int cnt) { // form a batch of cnt <= 64 candidates and issue its gathers
        a_valid = false;
        a_amb = false;
        a_pos = 0x7FFFFFFF;
        if (lane < cnt) {
            const int ent = ring[(head + lane) & (kRing - 1)];
            a_pos = ent & 0x7FFFFFFF;
            if (a_pos < n_ord) {
                a_valid = true;
                a_amb = ent < 0;
                a_pay = pay[a_pos];
                if (a_amb) a_end = end[a_pos];
            }
        }
        a_have = true;
        head = (head + cnt) & (kRing - 1);
        fill -= cnt;
    };

    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr unsigned kTop = sizeof(KT) == 2 ? 0x80008000u : 0x80808080u;
    const unsigned nk_ge = sizeof(KT) == 2 ? (now_key | (now_key << 16)) : now_key * 0x01010101u;
    const unsigned nk1 = now_key + 1u; // <= 0x80 / 0x8000: still no borrow out of a byte / half
    const unsigned nk_gt = sizeof(KT) == 2 ? (nk1 | (nk1 << 16)) : nk1 * 0x01010101u;
    auto row_bits = [&](unsigned g0, unsigned g1, unsigned g2, unsigned g3) -> unsigned { // flags (top bit per key) -> one bit per row, row order
        if constexpr (sizeof(KT) == 1) {
            auto nib = [](unsigned g) { return ((((g >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; };
            return nib(g0) | (nib(g1) << 4) | (nib(g2) << 8) | (nib(g3) << 12);
        } else {
            auto two = [](unsigned g) { return ((g >> 15) & 1u) | ((g >> 30) & 2u); };
            return two(g0) | (two(g1) << 2) | (two(g2) << 4) | (two(g3) << 6);
        }
    };

    const long long W = (long long)gridDim.x * 4;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    for (long long cb = gw; cb < n_chunks; cb += W * kUnroll) {
        u4_t kv[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const long long ch = cb + (long long)j * W;
            kv[j] = (u4_t){0u, 0u, 0u, 0u};
            if (ch < n_chunks) {
                if constexpr (sizeof(KT) == 1) {
                    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
                    const u2_t h = __builtin_nontemporal_load(reinterpret_cast<const u2_t*>(key + (ch << kChunkShift) + kPerLane * lane));
                    kv[j].x = h.x;
                    kv[j].y = h.y;
                } else kv[j] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(key + (ch << kChunkShift) + kPerLane * lane));
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const long long ch = cb + (long long)j * W;
            if (ch >= n_chunks) continue; // wave-uniform
            const unsigned ge = row_bits(((kv[j].x | kTop) - nk_ge) & kTop, ((kv[j].y | kTop) - nk_ge) & kTop,
                                         ((kv[j].z | kTop) - nk_ge) & kTop, ((kv[j].w | kTop) - nk_ge) & kTop) & 0xFFu; // 8 rows per lane
            const unsigned gt = row_bits(((kv[j].x | kTop) - nk_gt) & kTop, ((kv[j].y | kTop) - nk_gt) & kTop,
                                         ((kv[j].z | kTop) - nk_gt) & kTop, ((kv[j].w | kTop) - nk_gt) & kTop);
            const int cnt = __popc(ge);
            const int incl = wave_incl_scan_i32(cnt, lane);
            const int total = __shfl(incl, 63, kWave);
            if (total == 0) continue;
            chunk_max = max(chunk_max, total);
            int w = head + fill + incl - cnt;
            const int p0 = (int)(ch << kChunkShift) + kPerLane * lane;
            unsigned m = ge;
            while (m) {
                const int b = __ffs((int)m) - 1;
                ring[w & (kRing - 1)] = (p0 + b) | (((gt >> b) & 1u) ? 0 : (int)0x80000000);
                ++w;
                m &= m - 1;
            }
            fill += total;
            __builtin_amdgcn_wave_barrier();
            while (fill >= kWave) {
                step_b();
                step_a(kWave);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (fill > 0) {
        step_b();
        step_a(fill);
    }
    step_b();
    if (cur_chunk >= 0 && lane == 0) unit_count[cur_chunk] = cur_cnt;
    if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
    if (lane == 0 && namb) atomicAdd(&blk_amb, namb);
    if (lane == 0 && ncand) atomicAdd(&blk_cand, ncand);
    if (lane == 0 && chunk_max) atomicMax(&blk_chunk_max, chunk_max);
    __syncthreads();
    if (threadIdx.x == 0) add_row_stats(summary, blk_live, blk_amb, (int)blockIdx.x, blk_cand, blk_chunk_max);
}

// ------------------------------------------------------------------------------------------------ prefix

// Exclusive prefix of the unit counts, in two levels, one launch: block g scans the counts of group g (1024 units: four per
// thread) into unit_local[] and reports the group's sum; the block that finishes LAST (a counter, not a wait) scans the group
// sums into group_base[] and closes the list with M.  base(unit) = group_base[unit >> 10] + unit_local[unit].
constexpr int kOrdGroupShift = 10;
constexpr int kOrdGroup = 1 << kOrdGroupShift;

// blockIdx.y = query of a batch (strides 0 and one row of blocks for a single scan): every query has its own counts, prefixes,
// counter and summary
__global__ __launch_bounds__(256) void k_ord_prefix(const int* __restrict__ unit_count, long long n_units, int* __restrict__ unit_local,
                                                    long long* __restrict__ group_sum, long long* __restrict__ group_base,
                                                    OrdCtl* __restrict__ ctl, Summary* __restrict__ summary, long long unit_stride,
                                                    long long group_stride, long long sum_stride_bytes)
{
    __shared__ int wsum[4];
    __shared__ long long wsum64[4];
    __shared__ bool is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unit_count += (long long)blockIdx.y * unit_stride;
    unit_local += (long long)blockIdx.y * unit_stride;
    group_sum += (long long)blockIdx.y * group_stride;
    group_base += (long long)blockIdx.y * group_stride;
    summary = reinterpret_cast<Summary*>(reinterpret_cast<char*>(summary) + (long long)blockIdx.y * sum_stride_bytes);
    const long long n_groups = (n_units + kOrdGroup - 1) >> kOrdGroupShift;
    for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const long long u = (g << kOrdGroupShift) + 4 * (long long)threadIdx.x; // unit_count is padded to whole groups of zeros
        const int4 c = *reinterpret_cast<const int4*>(unit_count + u);
        const int mine = c.x + c.y + c.z + c.w;
        const int incl = wave_incl_scan_i32(mine, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int sw = wsum[w];
            if (w < wave) wbase += sw;
            total += sw;
        }
        int4 o;
        o.x = wbase + incl - mine;
        o.y = o.x + c.x;
        o.z = o.y + c.y;
        o.w = o.z + c.z;
        *reinterpret_cast<int4*>(unit_local + u) = o;
        if (threadIdx.x == 0) group_sum[g] = total;
        __syncthreads();
    }
    if (ctl == nullptr) return; // two-kernel form (batches: thousands of blocks, a fence each would cost more than a launch)
    ctl += blockIdx.y;
    if (threadIdx.x == 0) {
        __threadfence();
        is_last = atomicAdd(&ctl->done_prefix, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    long long carry = 0;
    for (long long g0 = 0; g0 < n_groups; g0 += 256) {
        const long long g = g0 + threadIdx.x;
        const long long v = g < n_groups ? __hip_atomic_load(&group_sum[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const long long incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum64[wave] = incl;
        __syncthreads();
        long long wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const long long sw = wsum64[w];
            if (w < wave) wbase += sw;
            total += sw;
        }
        if (g < n_groups) group_base[g] = carry + wbase + incl - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        group_base[n_groups] = carry;
        summary->m = (unsigned long long)carry;
        ctl->done_prefix = 0;
    }
}

// second kernel of the two-kernel form: block q scans query q's group sums

// ------------------------------------------------------------------------------------------------ emit: row list, offsets, counts, summary

// One launch, two kinds of block.  Blocks [0, copy_blocks) write the row list:
//   dense   block per tile: the tile's staged row ids are contiguous, so is their place in idx
//   keyed   thread per output element k: its unit is found by a two-level search (group bases, then the group's
//           unit_local[]) — balanced whatever the units hold (a head user's 80 chunks of 500 are 41 k elements like any
//           others); the staged entry is a position, the row id comes from the run's record
// Blocks [copy_blocks, ...) look after the users: thread t of user-block b has user b * 255 + t (the last thread's user is
// the next block's first: its rank closes the block's last count); offsets[u] = selected positions before uoff[u],
// u = n_users closes the list with M.
template <bool KEYED>
__global__ __launch_bounds__(256) void k_ord_emit(const long long* __restrict__ uoff, int n_users, long long n_ord, int unit_shift,
                                                  const int* __restrict__ unit_count, const int* __restrict__ unit_local,
                                                  const long long* __restrict__ group_base, long long n_units,
                                                  const unsigned int* __restrict__ stage, const OrdRec* __restrict__ pay,
                                                  const unsigned long long* __restrict__ tile_ballot,
                                                  const unsigned int* __restrict__ tile_prefix, int* __restrict__ out_idx,
                                                  long long* __restrict__ offsets, int* __restrict__ counts_ord, int copy_blocks,
                                                  Summary* __restrict__ summary, int* __restrict__ zero_counts, long long zero_n)
{
    __shared__ long long soff[256];
    __shared__ long long gb[1024];
    __shared__ int wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_groups = (n_units + kOrdGroup - 1) >> kOrdGroupShift;
    // the other unit-count buffer starts the next ordered scan clean
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long long)gridDim.x * 256) zero_counts[i] = 0;
    if ((int)blockIdx.x < copy_blocks) {
        if constexpr (KEYED) {
            const long long m = group_base[n_groups];
            const bool in_lds = n_groups < 1024;
            if (in_lds)
                for (int i = threadIdx.x; i <= (int)n_groups; i += 256) gb[i] = group_base[i];
            __syncthreads();
            for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < m; k += (long long)copy_blocks * 256) {
                long long lo = 0, hi = n_groups - 1; // the largest g with group_base[g] <= k: never an empty group
                while (lo < hi) {
                    const long long mid = (lo + hi + 1) >> 1;
                    if ((in_lds ? gb[mid] : group_base[mid]) <= k) lo = mid;
                    else hi = mid - 1;
                }
                const int x = (int)(k - (in_lds ? gb[lo] : group_base[lo]));
                const int* loc = unit_local + (lo << kOrdGroupShift);
                int a = 0, b = kOrdGroup - 1;        // the largest j with loc[j] <= x: never an empty unit
                while (a < b) {
                    const int mid = (a + b + 1) >> 1;
                    if (loc[mid] <= x) a = mid;
                    else b = mid - 1;
                }
                const long long unit = (lo << kOrdGroupShift) + a;
                const unsigned p = stage[(unit << unit_shift) + (x - loc[a])];
                out_idx[k] = pay[p].row;
            }
        } else {
            for (long long t = blockIdx.x; t < n_units; t += copy_blocks) {
                const int cnt = unit_count[t];
                const long long base = group_base[t >> kOrdGroupShift] + unit_local[t];
                const unsigned int* src = stage + (t << kOrdTileShift);
                for (int j = threadIdx.x; j < cnt; j += 256) out_idx[base + j] = (int)src[j];
            }
        }
    } else {
        const long long u = (long long)((int)blockIdx.x - copy_blocks) * 255 + threadIdx.x;
        long long my = 0;
        if (u <= n_users) {
            const long long q = uoff[u];
            if (q >= n_ord) my = group_base[n_groups];
            else {
                const long long unit = q >> unit_shift;
                my = group_base[unit >> kOrdGroupShift] + unit_local[unit];
                if constexpr (KEYED) {
                    const unsigned int* s = stage + (unit << unit_shift);
                    int lo = 0, hi = unit_count[unit];
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (s[mid] < (unsigned)q) lo = mid + 1;
                        else hi = mid;
                    }
                    my += lo;
                } else {
                    const int in = (int)(q & (kOrdTile - 1));
                    const long long sl = unit * kOrdSlices + (in >> 6);
                    my += tile_prefix[sl] + __popcll(tile_ballot[sl] & ((1ull << (in & 63)) - 1ull));
                }
            }
            offsets[u] = my;
        }
        soff[threadIdx.x] = my;
        __syncthreads();
        int cnt = 0;
        if (threadIdx.x < 255 && u < n_users) {
            cnt = (int)(soff[threadIdx.x + 1] - my);
            counts_ord[u] = cnt;
        }
        int mx = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, kWave));
        if (lane == 0) wmax[wave] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int bm = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
            if (bm > 0) atomicMax(&stat_slots(summary)[blockIdx.x & (kStatSlots - 1)].max_count, (unsigned long long)bm); // not one address for all blocks
        }
    }
}

// One wave behind the emit kernel: the row statistics of the table pass, M and the largest bucket -> mapped host memory, seq
// last.  (A "last block publishes" tail inside k_ord_emit would cost every one of its thousands of blocks a device-scope
// fence — an L2 write-back each, right after the row list was written — and an atomic on one address: measured 0.25 ms on
// the dense query.  A kernel boundary orders the same thing for a few microseconds.)
__global__ __launch_bounds__(64) void k_ord_publish(Summary* __restrict__ summary, HostSummary* __restrict__ host, unsigned long long seq,
                                                    long long sum_stride_bytes)
{
    const int lane = threadIdx.x;
    summary = reinterpret_cast<Summary*>(reinterpret_cast<char*>(summary) + (long long)blockIdx.x * sum_stride_bytes); // a batch: one block per query
    host += blockIdx.x;
    unsigned long long live = 0, amb = 0, cand = 0;
    unsigned int chunk_max = 0;
    sum_row_stats(summary, lane, live, amb, &cand, &chunk_max);
    unsigned long long slot_max = stat_slots(summary)[lane].max_count; // the batched emit's per-block maxima
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(slot_max, o, kWave);
        slot_max = other > slot_max ? other : slot_max;
    }
    { // this slot's statistics start the next ordered scan clean
        StatSlot* slot = stat_slots(summary) + lane;
        slot->max_count = 0;
        slot->live = 0;
        slot->amb = 0;
        slot->cand = 0;
        slot->chunk_max = 0;
    }
    if (lane == 0) {
        Summary out{};
        out.m = summary->m;
        out.max_count = max(summary->max_count, (unsigned int)slot_max);
        out.live = live;
        out.amb = amb;
        out.cand = cand;
        out.chunk_max = chunk_max;
        summary->max_count = 0; // m stays: the pack kernel reads it from here, and the next prefix overwrites it
        host->s = out;
        __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------ batched form: Q queries, one pass over the key column

// A table whose users are skewed cannot run the general batched pass (a head user's rows do not fit any union bucket), so
// its batches used to run as Q single scans, each streaming the key column again.  On the run a batch of up to 64 queries
// is ONE pass: the candidates are the positions whose key reaches the smallest key(now) of the batch, every candidate's
// 64-bit query mask comes from the three lookups of BatchTables (one 16-byte gather, `end` only where the key cannot decide),
// and a position that any query selects is staged once, in position order.  The result is the union, as on the general pass:
//   k_ord_batch_scan_t   key stream -> union staging records + per-chunk union counts
//   k_ord_prefix         exclusive prefix of the per-chunk counts (one launch; it also yields the per-unit bases)
//   k_ord_union_emit     copy role: staging record -> urows / umlo / umhi at its union position; user role: uoff from the
//                        run's per-user offsets, per-query totals (bit-sliced counting), the largest union bucket
//   k_ord_union_publish  the batch's summary + per-query totals to mapped host memory
// Per-query lists, where a caller asks for them, are cut from the union by the kernels the general pass uses (k_mat_*).
// A union staging record lives in its chunk's 512 slots, so its position is the chunk + 9 bits; the rest of the word carries the
// three table ranks the record's 64-bit query mask is rebuilt from (BatchTables): [place : 9 | liveness rank : 7 | window rank :
// 7 | discipline : 6].  The row id rides in the record: the emit reads the row where it reads the ranks, not through a second,
// dependent gather at the position.
struct alignas(8) OrdUnion {
    int row;
    unsigned qsub;
};
constexpr unsigned kOrdSubMask = 511u; // k_ord_batch_scan's chunk: 512 positions for both key widths

// A chunk is 512 positions for both key widths: 8 one-byte keys (one 8-byte load) or 8 two-byte keys (one 16-byte load) per
// lane.  The candidate ring then takes 5 / 6 KiB per wave instead of 10, and two to three times as many waves fit a CU —
// the pass is latency-bound (34 chunks and a handful of candidate batches per wave), so that is what counts: 141 -> 9x us.
// The table-lookup form (round 3): up to 64 queries per pass.  Identical streaming, candidate ring and position-ordered staging;
// the per-candidate loop over the queries is replaced by the three lookups of BatchTables (pie_kernels.h) and the staging word
// carries [place in chunk : 9 | liveness rank : 7 | window rank : 7 | discipline : 6] instead of 16 query bits.
template <class KT, int UNROLL = 4>
__global__ __launch_bounds__(256) void k_ord_batch_scan_t(const OrdRec* __restrict__ pay, const long long* __restrict__ end,
                                                        const KT* __restrict__ key, long long n_ord, long long n_chunks, unsigned min_key,
                                                        BatchTables tabs, OrdUnion* __restrict__ ustage, int* __restrict__ ucount,
                                                        Summary* __restrict__ summary)
{
    constexpr int kPerLane = 8;
    constexpr int kChunk = kPerLane * kWave;
    constexpr int kChunkShift = 9;
    constexpr int kRing = 2 * kChunk;
    constexpr int kUnroll = UNROLL;
    __shared__ int ring_s[4][kRing];
    __shared__ KT ringk_s[4][kRing];
    __shared__ int blk_cand, blk_chunk_max;
    __shared__ BatchTables tab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { blk_cand = 0; blk_chunk_max = 0; }
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(&tabs);
        unsigned* dst = reinterpret_cast<unsigned*>(&tab);
        for (int i = threadIdx.x; i < (int)(sizeof(BatchTables) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    int* ring = ring_s[wave];
    KT* ringk = ringk_s[wave];
    int head = 0, fill = 0, ncand = 0, chunk_max = 0;
    int cur_chunk = -1, cur_cnt = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    bool a_have = false, a_valid = false, a_amb = false;
    int a_r = 0; // queries whose key(now) lies below the candidate's key (see BatchTables)
    int a_pos = 0x7FFFFFFF;
    unsigned a_key = 0;
    OrdRec a_pay;
    a_pay.start = 0; a_pay.row = 0; a_pay.disc = -1;
    long long a_end = 0;

    auto step_b = [&]() {
        if (!a_have) return;
        a_have = false;
        // the Q predicates by table lookup (BatchTables): liveness rank (by the 8-byte `end` where a key cannot decide), window
        // rank, discipline; the record keeps the three ranks — 7 + 7 + 6 bits beside the 9 of its place in the chunk — and the
        // emit rebuilds the 64-bit query mask from them
        unsigned code = 0;
        bool sel = false;
        if (a_valid) {
            const int dv = a_pay.disc;
            int r = a_r;
            if (a_amb) r = rank_in_64<true>(tab.now, a_end);
            const int w = rank_in_64<false>(tab.cutoff, a_pay.start);
            if ((unsigned)dv < 64u) {
                sel = (tab.live[r] & tab.win[w] & tab.disc[dv]) != 0ull;
                code = ((unsigned)r << 9) | ((unsigned)w << 16) | ((unsigned)dv << 23);
            }
        }
        ncand += __popcll(__ballot(a_valid));
        const unsigned long long sb = __ballot(sel);
        if (sb == 0) return;
        const int ch = a_pos >> kChunkShift;
        const int ch_below = __shfl_up(ch, 1, kWave);
        const unsigned long long hb = __ballot(lane == 0 || ch != ch_below);
        const int first = 63 - __clzll((long long)(hb & le));
        int rank = __popcll(sb & lt & ~((1ull << first) - 1ull));
        if (ch == cur_chunk) rank += cur_cnt;
        const unsigned long long ha = hb & ~le;
        const int seg_end = ha ? __ffsll((long long)ha) - 1 : 64;
        const unsigned long long upto = seg_end == 64 ? ~0ull : ((1ull << seg_end) - 1ull);
        const bool last_of_seg = sel && (sb & ~le & upto) == 0;
        const int last = 63 - __clzll((long long)sb);
        const int new_chunk = __shfl(ch, last, kWave);
        const int new_cnt = __shfl(rank, last, kWave) + 1;
        if (sel) {
            OrdUnion r;
            r.row = a_pay.row;
            r.qsub = code | ((unsigned)a_pos & kOrdSubMask);
            ustage[((long long)ch << kChunkShift) + rank] = r;
            if (last_of_seg && ch != new_chunk) ucount[ch] = rank + 1;
        }
        if (cur_chunk >= 0 && cur_chunk != new_chunk && __ballot(sel && ch == cur_chunk) == 0 && lane == 0) ucount[cur_chunk] = cur_cnt;
        cur_chunk = new_chunk;
        cur_cnt = new_cnt;
    };
    auto step_a = [&](int cnt) {
        a_valid = false;
        a_amb = false;
        a_pos = 0x7FFFFFFF;
        if (lane < cnt) {
            const int slot = (head + lane) & (kRing - 1);
            a_pos = ring[slot];
            a_key = ringk[slot];
            if (a_pos < n_ord) {
                a_valid = true;
                a_pay = pay[a_pos];
                a_r = rank_in_64<true>(tab.nk, a_key);
                a_amb = a_r < kBatchMax && tab.nk[a_r < kBatchMax ? a_r : 0] == a_key;
                if (a_amb) a_end = end[a_pos];
            }
        }
        a_have = true;
        head = (head + cnt) & (kRing - 1);
        fill -= cnt;
    };

    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr unsigned kTop = sizeof(KT) == 2 ? 0x80008000u : 0x80808080u;
    const unsigned mk = min_key;
    const unsigned nk_ge = sizeof(KT) == 2 ? (mk | (mk << 16)) : mk * 0x01010101u;
    auto row_bits = [&](unsigned g0, unsigned g1, unsigned g2, unsigned g3) -> unsigned {
        if constexpr (sizeof(KT) == 1) {
            auto nib = [](unsigned g) { return ((((g >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; };
            return nib(g0) | (nib(g1) << 4) | (nib(g2) << 8) | (nib(g3) << 12);
        } else {
            auto two = [](unsigned g) { return ((g >> 15) & 1u) | ((g >> 30) & 2u); };
            return two(g0) | (two(g1) << 2) | (two(g2) << 4) | (two(g3) << 6);
        }
    };
    const long long W = (long long)gridDim.x * 4;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    for (long long cb = gw; cb < n_chunks; cb += W * kUnroll) {
        u4_t kv[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const long long ch = cb + (long long)j * W;
            kv[j] = (u4_t){0u, 0u, 0u, 0u};
            if (ch < n_chunks) {
                if constexpr (sizeof(KT) == 1) {
                    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
                    const u2_t h = __builtin_nontemporal_load(reinterpret_cast<const u2_t*>(key + (ch << kChunkShift) + kPerLane * lane));
                    kv[j].x = h.x;
                    kv[j].y = h.y;
                } else kv[j] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(key + (ch << kChunkShift) + kPerLane * lane));
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const long long ch = cb + (long long)j * W;
            if (ch >= n_chunks) continue;
            const unsigned ge = row_bits(((kv[j].x | kTop) - nk_ge) & kTop, ((kv[j].y | kTop) - nk_ge) & kTop,
                                         ((kv[j].z | kTop) - nk_ge) & kTop, ((kv[j].w | kTop) - nk_ge) & kTop) & 0xFFu; // 8 rows per lane
            const int cnt = __popc(ge);
            const int incl = wave_incl_scan_i32(cnt, lane);
            const int total = __shfl(incl, 63, kWave);
            if (total == 0) continue;
            chunk_max = max(chunk_max, total);
            int w = head + fill + incl - cnt;
            const int p0 = (int)(ch << kChunkShift) + kPerLane * lane;
            unsigned m = ge;
            while (m) {
                const int b = __ffs((int)m) - 1;
                unsigned kq;
                if constexpr (sizeof(KT) == 1) {
                    const unsigned word = (b >> 2) == 0 ? kv[j].x : (b >> 2) == 1 ? kv[j].y : (b >> 2) == 2 ? kv[j].z : kv[j].w;
                    kq = (word >> (8 * (b & 3))) & 0xFFu;
                } else {
                    const unsigned word = (b >> 1) == 0 ? kv[j].x : (b >> 1) == 1 ? kv[j].y : (b >> 1) == 2 ? kv[j].z : kv[j].w;
                    kq = (word >> (16 * (b & 1))) & 0xFFFFu;
                }
                ring[w & (kRing - 1)] = p0 + b;
                ringk[w & (kRing - 1)] = (KT)kq;
                ++w;
                m &= m - 1;
            }
            fill += total;
            __builtin_amdgcn_wave_barrier();
            while (fill >= kWave) {
                step_b();
                step_a(kWave);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (fill > 0) {
        step_b();
        step_a(fill);
    }
    step_b();
    if (cur_chunk >= 0 && lane == 0) ucount[cur_chunk] = cur_cnt;
    if (lane == 0 && ncand) atomicAdd(&blk_cand, ncand);
    if (lane == 0 && chunk_max) atomicMax(&blk_chunk_max, chunk_max);
    __syncthreads();
    if (threadIdx.x == 0) add_row_stats(summary, 0, 0, (int)blockIdx.x, blk_cand, blk_chunk_max);
}

constexpr int kOrdChunkHeavy = 8; // a chunk with more staged records than this is copied by a whole wave

// ---- the UNION form of a batch on the run (round 3; batches of <= 16 queries).  The staging records of k_ord_batch_scan — one per
// position that ANY query selects, in position order, with the query bits — ARE the batch's union in (user, start, row) order.
// So the chain behind the scan no longer works per query (round 2: a per-query count kernel, Q prefix scans, a per-query
// compaction by ballot in the emit, Q summaries — six launches, 0.133 ms for 16 queries on the Zipf table):
//   k_ord_prefix        ONE exclusive prefix, over the chunks' union counts (the single-scan form: one launch)
//   k_ord_union_emit    copy role: every chunk's records -> urows / umlo behind the chunk's prefix (a lane per light chunk, heavy
//                       chunks by whole waves, neighbours on different waves), per-query totals on the way;
//                       user role: uoff[u] = prefix of the chunk that holds the user's segment start + the records before it
//   k_ord_union_publish summary + per-query totals -> the batch's mapped host block
// and the result is the same union (uoff / urows / umlo) the general batched pass returns: per-query lists, messages and request
// fetches are produced from it by the same kernels.
// bit-sliced count of up to 8 mask words: plane i, bit q = bit i of "how many of the 8 words have bit q"
__device__ __forceinline__ void csa8(const unsigned (&m)[8], unsigned (&pl)[4])
{
    auto fa = [](unsigned a, unsigned b, unsigned c, unsigned& carry) { const unsigned x = a ^ b; carry = (a & b) | (c & x); return x ^ c; };
    unsigned c1, c2, c3, c4, c5, c6;
    const unsigned s1 = fa(m[0], m[1], m[2], c1), s2 = fa(m[3], m[4], m[5], c2);
    const unsigned s3 = m[6] ^ m[7];
    c3 = m[6] & m[7];
    pl[0] = fa(s1, s2, s3, c4);
    const unsigned tw = fa(c1, c2, c3, c5);
    pl[1] = tw ^ c4;
    c6 = tw & c4;
    pl[2] = c5 ^ c6;
    pl[3] = c5 & c6;
}

// HI: the batch holds more than 32 queries (a second mask word per union row)
template <bool HI>
__global__ __launch_bounds__(256) void k_ord_union_emit(const long long* __restrict__ uoff_run, int n_users, long long n_ord, int chunk_shift,
                                                        long long n_chunks, int n_q, BatchTables tabs, const OrdUnion* __restrict__ ustage,
                                                        const int* __restrict__ ucount, const int* __restrict__ unit_local,
                                                        const long long* __restrict__ group_base, long long* __restrict__ uoff_out,
                                                        int* __restrict__ urows, unsigned int* __restrict__ umlo, unsigned int* __restrict__ umhi,
                                                        long long ucap, int copy_blocks, Summary* __restrict__ summary,
                                                        unsigned int* __restrict__ mq_slots, int* __restrict__ zero_counts, long long zero_n)
{
    const int user_blocks = (int)gridDim.x - copy_blocks;
    __shared__ long long soff[256];
    __shared__ int wmax[4];
    __shared__ BatchTables tab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_groups = (n_chunks + kOrdGroup - 1) >> kOrdGroupShift;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long long)gridDim.x * 256) zero_counts[i] = 0;
    if ((int)blockIdx.x >= user_blocks) {
        {
            const unsigned* src = reinterpret_cast<const unsigned*>(&tabs);
            unsigned* dst = reinterpret_cast<unsigned*>(&tab);
            for (int i = threadIdx.x; i < (int)(sizeof(BatchTables) / 4); i += 256) dst[i] = src[i];
        }
        __syncthreads();
        // the 64-bit query mask of a staging record, from its three ranks
        auto mask_of = [&](unsigned qsub) { return tab.live[(qsub >> 9) & 127u] & tab.win[(qsub >> 16) & 127u] & tab.disc[(qsub >> 23) & 63u]; };
        const long long n_waves = (long long)copy_blocks * 4, wv = (long long)((int)blockIdx.x - user_blocks) * 4 + wave;
        const int nq_lo = n_q < 32 ? n_q : 32;
        unsigned acc = 0; // lane q holds query q's count
        auto count_planes = [&](const unsigned (&pl)[4], int q0, int q1) { // four ballots per query: the wave's total of bit q - q0 of the planes
#pragma unroll 4
            for (int q = q0; q < q1; ++q) {
                const int sh = q - q0;
                const unsigned c = (unsigned)__popcll(__ballot((pl[0] >> sh) & 1u)) + 2u * (unsigned)__popcll(__ballot((pl[1] >> sh) & 1u)) +
                                   4u * (unsigned)__popcll(__ballot((pl[2] >> sh) & 1u)) + 8u * (unsigned)__popcll(__ballot((pl[3] >> sh) & 1u));
                if (lane == q) acc += c;
            }
        };
        // pass 1: a lane per chunk, 64 consecutive chunks per wave step: a chunk holds a handful of records
        for (long long base = wv << 6; base < n_chunks; base += n_waves << 6) {
            const long long ch = base + lane;
            const int cnt = ch < n_chunks ? ucount[ch] : 0;
            unsigned mlo[8], mhi[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { mlo[j] = 0; mhi[j] = 0; }
            if (cnt > 0 && cnt <= kOrdChunkHeavy) {
                const OrdUnion* rec = ustage + (ch << chunk_shift);
                const long long dst = group_base[ch >> kOrdGroupShift] + unit_local[ch];
#pragma unroll
                for (int j = 0; j < kOrdChunkHeavy; ++j) {
                    if (j < cnt) {
                        const OrdUnion r = rec[j];
                        const unsigned long long m = mask_of(r.qsub);
                        mlo[j] = (unsigned)m;
                        mhi[j] = (unsigned)(m >> 32);
                        if (dst + j < ucap) {
                            urows[dst + j] = r.row;
                            umlo[dst + j] = (unsigned)m;
                            if constexpr (HI) umhi[dst + j] = (unsigned)(m >> 32);
                        }
                    }
                }
            }
            unsigned pl[4];
            csa8(mlo, pl);
            count_planes(pl, 0, nq_lo);
            if constexpr (HI) {
                csa8(mhi, pl);
                count_planes(pl, 32, n_q);
            }
        }
        // pass 2: the heavy chunks (a popular user's live rows: whole runs of full chunks), chunk ch by wave ch mod n_waves
        for (long long c0 = wv; c0 < n_chunks; c0 += n_waves << 6) {
            const long long ch = c0 + (long long)lane * n_waves;
            const int cnt = ch < n_chunks ? ucount[ch] : 0;
            unsigned long long todo = __ballot(cnt > kOrdChunkHeavy);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const long long lch = c0 + (long long)leader * n_waves;
                const int lcnt = __shfl(cnt, leader, kWave);
                const OrdUnion* lrec = ustage + (lch << chunk_shift);
                const long long ldst = group_base[lch >> kOrdGroupShift] + unit_local[lch];
                for (int j0 = 0; j0 < lcnt; j0 += 64) {
                    unsigned long long m = 0;
                    if (j0 + lane < lcnt) {
                        const OrdUnion r = lrec[j0 + lane];
                        m = mask_of(r.qsub);
                        if (ldst + j0 + lane < ucap) {
                            urows[ldst + j0 + lane] = r.row;
                            umlo[ldst + j0 + lane] = (unsigned)m;
                            if constexpr (HI) umhi[ldst + j0 + lane] = (unsigned)(m >> 32);
                        }
                    }
#pragma unroll 4
                    for (int q = 0; q < nq_lo; ++q) {
                        const unsigned c = (unsigned)__popcll(__ballot(((unsigned)m >> q) & 1u));
                        if (lane == q) acc += c;
                    }
                    if constexpr (HI) {
#pragma unroll 4
                        for (int q = 32; q < n_q; ++q) {
                            const unsigned c = (unsigned)__popcll(__ballot(((unsigned)(m >> 32) >> (q - 32)) & 1u));
                            if (lane == q) acc += c;
                        }
                    }
                }
                todo &= todo - 1;
            }
        }
        if (lane < n_q && acc) atomicAdd(&mq_slots[(blockIdx.x & (kMqSlots - 1)) * kBatchMax + lane], acc);
        return;
    }
    // user role: 255 users per block (thread 255 holds the next block's first user, for the bucket sizes)
    const long long u = (long long)blockIdx.x * 255 + threadIdx.x;
    long long my = 0;
    if (u <= n_users) {
        const long long qpos = uoff_run[u];
        if (qpos >= n_ord) my = group_base[n_groups];
        else {
            const long long ch = qpos >> chunk_shift;
            const OrdUnion* rec = ustage + (ch << chunk_shift);
            int lo = 0, hi = ucount[ch];
            while (lo < hi) { // staged in position order: the records before my segment start
                const int mid = (lo + hi) >> 1;
                if ((rec[mid].qsub & kOrdSubMask) < ((unsigned)qpos & kOrdSubMask)) lo = mid + 1;
                else hi = mid;
            }
            my = group_base[ch >> kOrdGroupShift] + unit_local[ch] + lo;
        }
        uoff_out[u] = my;
    }
    soff[threadIdx.x] = my;
    __syncthreads();
    int cnt = 0;
    if (threadIdx.x < 255 && u < n_users) cnt = (int)(soff[threadIdx.x + 1] - soff[threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, __shfl_xor(cnt, o, kWave));
    if (lane == 0) wmax[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int bm = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (bm > 0) atomicMax(&stat_slots(summary)[blockIdx.x & (kStatSlots - 1)].max_count, (unsigned long long)bm);
    }
}

// one block: the batch's summary (m = union rows, the largest union bucket, the pass's row statistics) and the per-query totals
// into mapped host memory, seq last; the counters it read start the next batch clean
__global__ __launch_bounds__(256) void k_ord_union_publish(Summary* __restrict__ summary, unsigned int* __restrict__ mq_slots, int n_q, long long ucap,
                                                           BatchHost* __restrict__ host, unsigned long long seq)
{
    __shared__ unsigned int s_mq[4][kBatchMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        unsigned v[kMqSlots / 4];
#pragma unroll
        for (int i = 0; i < kMqSlots / 4; ++i) {
            unsigned int* p = &mq_slots[(wave * (kMqSlots / 4) + i) * kBatchMax + lane];
            v[i] = lane < n_q ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (lane < n_q) *p = 0;
        }
        unsigned tot = 0;
#pragma unroll
        for (int i = 0; i < kMqSlots / 4; ++i) tot += v[i];
        s_mq[wave][lane] = tot;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    unsigned long long live = 0, amb = 0, cand = 0;
    unsigned int chunk_max = 0;
    sum_row_stats(summary, lane, live, amb, &cand, &chunk_max);
    unsigned long long slot_max = stat_slots(summary)[lane].max_count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(slot_max, o, kWave);
        slot_max = other > slot_max ? other : slot_max;
    }
    {
        StatSlot* slot = stat_slots(summary) + lane;
        slot->max_count = 0;
        slot->live = 0;
        slot->amb = 0;
        slot->cand = 0;
        slot->chunk_max = 0;
    }
    host->mq[lane] = (unsigned long long)s_mq[0][lane] + s_mq[1][lane] + s_mq[2][lane] + s_mq[3][lane];
    if (lane == 0) {
        Summary out{};
        out.m = summary->m;
        out.max_count = (unsigned int)slot_max;
        out.cand = cand;
        out.chunk_max = chunk_max;
        out.bad_rows = 0;
        out.n_over = (long long)out.m > ucap ? 1u : 0u; // the union outgrew the result arrays: the host reruns the queries
        host->s = out;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace pie
