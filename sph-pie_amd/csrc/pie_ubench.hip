// Measurement aid, not part of the scan library (its own shared object: libpie_ubench.so): the streaming-READ ceiling of the GPU
// the bench is running on, measured in the same process as the scan it is compared with.  MI355X boxes of one pool differ by
// 5 - 8 % in what an HBM-bound kernel reaches (the every-byte scan's table pass took 0.384 ms on one box
// and 0.421 ms on another, same binary), so "fraction of the 8 TB/s spec" moves with the box; "fraction of what a kernel that
// does nothing but read reaches on THIS box" does not.  The kernel below is that kernel: every lane streams 16-byte
// nontemporal loads (the scan's own load form) over one large buffer and folds them into a word nobody reads.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

namespace {

typedef unsigned u4_t __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u4_t* __restrict__ src, long long n16, unsigned* __restrict__ sink)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u4_t v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) v[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    for (; i < n16; i += stride) {
        const u4_t v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 63] = acc; // (never true for the fill pattern: the loads must still happen)
}

// every block owns ONE contiguous range (the scan's own split) instead of the grid-stride interleave above
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read_ranges(const u4_t* __restrict__ src, long long n16, unsigned* __restrict__ sink)
{
    const long long per = ((n16 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long b0 = (long long)blockIdx.x * per;
    long long b1 = b0 + per;
    if (b1 > n16) b1 = n16;
    unsigned acc = 0;
    long long i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
        u4_t v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) v[j] = __builtin_nontemporal_load(src + i + j * 256);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    for (; i < b1; i += 256) {
        const u4_t v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 63] = acc;
}

// the scan's own access pattern without the scan: the buffer read as FOUR columns (8, 8, 4, 4 bytes per row), every block a
// contiguous row range, per wave and step 128 rows = one 16-byte load from each 8-byte column and one 8-byte load from each
// 4-byte column per lane
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read_columns(const char* __restrict__ base, long long rows, unsigned* __restrict__ sink)
{
    const u4_t* c0 = reinterpret_cast<const u4_t*>(base);
    const u4_t* c1 = reinterpret_cast<const u4_t*>(base + 8 * rows);
    const u2_t* c2 = reinterpret_cast<const u2_t*>(base + 16 * rows);
    const u2_t* c3 = reinterpret_cast<const u2_t*>(base + 20 * rows);
    const long long pairs = rows / 2; // a lane's unit: two rows
    const long long per = ((pairs + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long b0 = (long long)blockIdx.x * per;
    long long b1 = b0 + per;
    if (b1 > pairs) b1 = pairs;
    unsigned acc = 0;
    long long i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
        u4_t v0[UNROLL], v1[UNROLL];
        u2_t v2[UNROLL], v3[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            v0[j] = __builtin_nontemporal_load(c0 + i + j * 256);
            v1[j] = __builtin_nontemporal_load(c1 + i + j * 256);
            v2[j] = __builtin_nontemporal_load(c2 + i + j * 256);
            v3[j] = __builtin_nontemporal_load(c3 + i + j * 256);
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= v0[j].x ^ v0[j].y ^ v0[j].z ^ v0[j].w ^ v1[j].x ^ v1[j].y ^ v1[j].z ^ v1[j].w ^ v2[j].x ^ v2[j].y ^ v3[j].x ^ v3[j].y;
    }
    for (; i < b1; i += 256) {
        const u4_t a = __builtin_nontemporal_load(c0 + i), b = __builtin_nontemporal_load(c1 + i);
        const u2_t c = __builtin_nontemporal_load(c2 + i), d = __builtin_nontemporal_load(c3 + i);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ d.x ^ d.y;
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 63] = acc;
}

// Experiments: the four-column read dressed step by step as the scan's streaming form (k_scan_compact) — where does the
// scan lose its 15 % against the plain read?
//   LEVEL 0  the scan's split: block = contiguous row range, the block's four waves take 512-row tiles in turn (UNROLL units of
//            128 rows each per tile), loads in the scan's order (end, start, disc, user)
//   LEVEL 1  + a block-wide barrier at the start and the end, 10 KB of LDS per block
//   LEVEL 2  + the scan's per-row work without its atomics: predicate, live-row ballots, neighbour statistic, one ballot per
//            64-row slice for the compaction
typedef long long ll2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

template <int UNROLL, int LEVEL>
__global__ __launch_bounds__(256) void k_read_like_scan(const long long* __restrict__ start, const long long* __restrict__ end,
                                                        const int* __restrict__ user, const int* __restrict__ disc, long long n,
                                                        long long rows_per_block, long long now, long long cutoff, unsigned long long mask,
                                                        unsigned* __restrict__ sink)
{
    __shared__ int lds[LEVEL >= 1 ? 2560 : 1];
    __shared__ int blk_live;
    constexpr int kUnit = 128, kTile = kUnit * UNROLL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LEVEL >= 1) {
        if (threadIdx.x == 0) { blk_live = 0; lds[0] = 0; }
        __syncthreads();
    }
    const long long c0 = (long long)blockIdx.x * rows_per_block;
    long long c1 = c0 + rows_per_block;
    if (c1 > n) c1 = n;
    unsigned acc = 0;
    int nlive = 0, ndup = 0, nsel = 0;
    for (long long t = c0 + (long long)wave * kTile; t + kTile <= c1; t += (long long)kTile * 4) {
        ll2_t s[UNROLL], e[UNROLL];
        i2_t u[UNROLL], d[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const long long r = t + j * kUnit + 2 * lane;
            e[j] = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(end + r));
            s[j] = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(start + r));
            d[j] = __builtin_nontemporal_load(reinterpret_cast<const i2_t*>(disc + r));
            u[j] = __builtin_nontemporal_load(reinterpret_cast<const i2_t*>(user + r));
        }
        if (LEVEL >= 2) {
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) {
                const bool p0 = e[j].x > now && s[j].x >= cutoff && (unsigned)d[j].x < 64u && ((mask >> (d[j].x & 63)) & 1ull);
                const bool p1 = e[j].y > now && s[j].y >= cutoff && (unsigned)d[j].y < 64u && ((mask >> (d[j].y & 63)) & 1ull);
                nlive += __popcll(__ballot(e[j].x > now)) + __popcll(__ballot(e[j].y > now));
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const bool p = k ? p1 : p0;
                    const int uu = k ? u[j].y : u[j].x;
                    const unsigned long long b = __ballot(p);
                    const unsigned long long below = b & ((1ull << lane) - 1ull);
                    const int prev = below ? 63 - __clzll((long long)below) : 0;
                    const int u_prev = __shfl(uu, prev, 64);
                    ndup += __popcll(__ballot(p && below != 0 && uu == u_prev));
                    nsel += __popcll(b);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < UNROLL; ++j)
                acc ^= (unsigned)e[j].x ^ (unsigned)e[j].y ^ (unsigned)s[j].x ^ (unsigned)s[j].y ^ (unsigned)d[j].x ^ (unsigned)d[j].y ^ (unsigned)u[j].x ^ (unsigned)u[j].y ^
                       (unsigned)(e[j].x >> 32) ^ (unsigned)(e[j].y >> 32) ^ (unsigned)(s[j].x >> 32) ^ (unsigned)(s[j].y >> 32);
        }
    }
    acc ^= (unsigned)(nlive * 3 + ndup * 5 + nsel * 7);
    if (LEVEL >= 1) {
        if (lane == 0 && nlive) atomicAdd(&blk_live, nlive);
        __syncthreads();
        if (threadIdx.x == 0 && blk_live == 0x7FFFFFF1) sink[1] = (unsigned)blk_live;
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 63] = acc;
}

} // namespace

extern "C" {

// forms of the dressed read above: ms_out[6] = LEVEL 0 / 1 / 2 with 12 288 blocks (the scan's grid), then LEVEL 2 with 6 144 and
// 24 576 blocks, then LEVEL 2 unroll 2 with 12 288
int pie_ubench_like_scan(int device, long long rows, int reps, double* ms_out)
{
    if (!ms_out || reps < 1 || reps > 64 || rows < 4096) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    rows &= ~4095LL;
    char* buf = nullptr;
    unsigned* sink = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    if (hipMalloc(&buf, (size_t)rows * 24) != hipSuccess || hipMalloc(&sink, 64 * 4) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        rc = -3;
    } else {
        (void)hipMemsetAsync(buf, 0x11, (size_t)rows * 24, s);
        (void)hipMemsetAsync(sink, 0, 64 * 4, s);
        const long long* c_start = reinterpret_cast<const long long*>(buf);
        const long long* c_end = reinterpret_cast<const long long*>(buf + 8 * rows);
        const int* c_user = reinterpret_cast<const int*>(buf + 16 * rows);
        const int* c_disc = reinterpret_cast<const int*>(buf + 20 * rows);
        const long long now = 0x1111111111111112LL, cutoff = 5;   // (every `end` is 0x1111...11: nothing is live, nothing selected)
        for (int form = 0; form < 6 && rc == 0; ++form) {
            const long long blocks_want = form == 3 ? 6144 : form == 4 ? 24576 : 12288;
            long long rpb = ((rows + blocks_want - 1) / blocks_want + 4095) / 4096 * 4096;
            const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
            float t[64];
            for (int r = -1; r < reps && rc == 0; ++r) {
                (void)hipEventRecord(e0, s);
                switch (form) {
                case 0: hipLaunchKernelGGL((k_read_like_scan<4, 0>), dim3(grid), dim3(256), 0, s, c_start, c_end, c_user, c_disc, rows, rpb, now, cutoff, 0x5555555555555555ull, sink); break;
                case 1: hipLaunchKernelGGL((k_read_like_scan<4, 1>), dim3(grid), dim3(256), 0, s, c_start, c_end, c_user, c_disc, rows, rpb, now, cutoff, 0x5555555555555555ull, sink); break;
                case 5: hipLaunchKernelGGL((k_read_like_scan<2, 2>), dim3(grid), dim3(256), 0, s, c_start, c_end, c_user, c_disc, rows, rpb, now, cutoff, 0x5555555555555555ull, sink); break;
                default: hipLaunchKernelGGL((k_read_like_scan<4, 2>), dim3(grid), dim3(256), 0, s, c_start, c_end, c_user, c_disc, rows, rpb, now, cutoff, 0x5555555555555555ull, sink); break;
                }
                (void)hipEventRecord(e1, s);
                if (hipEventSynchronize(e1) != hipSuccess) { rc = -4; break; }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 0) t[r] = ms;
            }
            if (rc) break;
            for (int a = 1; a < reps; ++a)
                for (int b = a; b > 0 && t[b - 1] > t[b]; --b) { const float x = t[b]; t[b] = t[b - 1]; t[b - 1] = x; }
            ms_out[form] = (double)t[reps / 2];
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (buf) (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    return rc;
}

// Streams `bytes` (rounded down to 16) `reps` times per form and reports, per form, the MEDIAN launch time in milliseconds
// (HIP events on the stream the kernel runs on).  forms: 0..7 = {grid-stride, contiguous ranges} x {16, 48 blocks per CU} x
// {unroll 4, 8} over ONE stream of bytes; 8..11 = the same bytes as four columns (8 + 8 + 4 + 4 bytes per row, bytes / 24 rows;
// the table's own layout), contiguous row ranges, {16, 48 blocks per CU} x {unroll 2, 4}; ms_out[12].  Returns 0, or a
// negative number when the device cannot be used / the buffer not allocated.
int pie_ubench_read_bw(int device, size_t bytes, int reps, double* ms_out)
{
    if (!ms_out || reps < 1 || reps > 64 || bytes < (1u << 20)) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    void* buf = nullptr;
    unsigned* sink = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64 * 4) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        rc = -3;
    } else {
        (void)hipMemsetAsync(buf, 0x5A, bytes, s);
        (void)hipMemsetAsync(sink, 0, 64 * 4, s);
        const long long n16 = (long long)(bytes / 16);
        const long long rows = (long long)(bytes / 24) & ~255LL;
        for (int form = 0; form < 12 && rc == 0; ++form) {
            const unsigned grid = (unsigned)cus * ((form & 2) ? 48u : 16u);
            float t[64];
            for (int r = -1; r < reps && rc == 0; ++r) { // r = -1: warm-up
                (void)hipEventRecord(e0, s);
                const u4_t* p = static_cast<const u4_t*>(buf);
                switch (form) {
                case 0: case 2: hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 1: case 3: hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 4: case 6: hipLaunchKernelGGL(k_read_ranges<4>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 5: case 7: hipLaunchKernelGGL(k_read_ranges<8>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 8: case 10: hipLaunchKernelGGL(k_read_columns<2>, dim3(grid), dim3(256), 0, s, static_cast<const char*>(buf), rows, sink); break;
                default: hipLaunchKernelGGL(k_read_columns<4>, dim3(grid), dim3(256), 0, s, static_cast<const char*>(buf), rows, sink); break;
                }
                (void)hipEventRecord(e1, s);
                if (hipEventSynchronize(e1) != hipSuccess) { rc = -4; break; }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 0) t[r] = ms;
            }
            if (rc) break;
            for (int a = 1; a < reps; ++a) // insertion sort: the median
                for (int b = a; b > 0 && t[b - 1] > t[b]; --b) { const float x = t[b]; t[b] = t[b - 1]; t[b - 1] = x; }
            ms_out[form] = (double)t[reps / 2];
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    if (buf) (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    return rc;
}

} // extern "C"
