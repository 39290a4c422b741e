// Micro-benchmark: rate of scattered returning integer atomics on a histogram of U counters, by memory scope.
//   agent scope      global_atomic_add ... sc1: performed where every XCD sees it (memory side)
//   workgroup scope  performed in the issuing XCD's L2 (NOT coherent across XCDs: only valid for per-XCD private data)
// usage: atomic_scope [U] [ops]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <int SCOPE, bool PER_XCD>
__global__ __launch_bounds__(256) void k_atomics(int* counts, int n_users, long long ops, int* sink)
{
    unsigned xcc = 0;
    if (PER_XCD) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    int* base = counts + (PER_XCD ? (long long)xcc * n_users : 0);
    int acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ops; i += (long long)gridDim.x * 256) {
        const int u = (int)(mix64(i * 0x9E3779B97F4A7C15ULL + 1) % (unsigned long long)n_users);
        acc += __hip_atomic_fetch_add(&base[u], 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0x7fffffff) *sink = acc;
}
int main(int argc, char** argv)
{
    const int U = argc > 1 ? atoi(argv[1]) : 100000;
    const long long ops = argc > 2 ? atoll(argv[2]) : 25000000LL;
    int *counts, *sink;
    hipMalloc(&counts, (size_t)U * 8 * 4);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto kernel) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipMemset(counts, 0, (size_t)U * 8 * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3(4096), dim3(256), 0, 0, counts, U, ops, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<int> h((size_t)U * 8);
        hipMemcpy(h.data(), counts, h.size() * 4, hipMemcpyDeviceToHost);
        long long tot = 0;
        for (int v : h) tot += v;
        printf("%-44s %8.3f ms  %7.1f G atomics/s  total %lld (%s)\n", name, best, ops / best / 1e6, tot, tot == ops ? "exact" : "LOST UPDATES");
    };
    run("agent scope, one histogram", k_atomics<__HIP_MEMORY_SCOPE_AGENT, false>);
    run("workgroup scope, one histogram (invalid)", k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, false>);
    run("agent scope, histogram per XCD", k_atomics<__HIP_MEMORY_SCOPE_AGENT, true>);
    run("workgroup scope, histogram per XCD", k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP, true>);
    run("wavefront scope, histogram per XCD", k_atomics<__HIP_MEMORY_SCOPE_WAVEFRONT, true>);
    return 0;
}
