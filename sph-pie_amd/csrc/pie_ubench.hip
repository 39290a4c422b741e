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

} // namespace

extern "C" {

// Streams `bytes` (rounded down to 16) `reps` times per form and reports, per form, the MEDIAN launch time in milliseconds
// (HIP events on the stream the kernel runs on).  forms: 8 of them — {grid-stride, contiguous ranges} x {16, 48 blocks per CU}
// x {unroll 4, 8}; ms_out[8].  Returns 0, or a negative number when the device cannot be used / the buffer not allocated.
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
        for (int form = 0; form < 8 && rc == 0; ++form) {
            const unsigned grid = (unsigned)cus * ((form & 2) ? 48u : 16u);
            float t[64];
            for (int r = -1; r < reps && rc == 0; ++r) { // r = -1: warm-up
                (void)hipEventRecord(e0, s);
                const u4_t* p = static_cast<const u4_t*>(buf);
                switch (form) {
                case 0: case 2: hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 1: case 3: hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                case 4: case 6: hipLaunchKernelGGL(k_read_ranges<4>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
                default: hipLaunchKernelGGL(k_read_ranges<8>, dim3(grid), dim3(256), 0, s, p, n16, sink); break;
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
