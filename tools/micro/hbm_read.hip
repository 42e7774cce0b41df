// Microbenchmark: what a bare read-only streaming kernel reaches on this MI355X (the practical HBM
// read ceiling the scan kernel is compared with in DESIGN.md).  16 B per lane, grid-stride, xor-fold
// so the loads cannot be eliminated; default-policy and non-temporal loads; buffer >> Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rd(const u32x4 *__restrict__ p, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 256 * (UNROLL - 1) < n; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <bool NT, int UNROLL>
int run(const char *name, const u32x4 *d, size_t n, uint32_t *out, int blocks) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rd<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, n, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((rd<NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, n, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s blocks=%6d  %.3f ms  %.1f GB/s\n", name, blocks, best, n * 16.0 / best / 1e6);
    return 0;
}
int main() {
    const size_t bytes = 12ull << 30;  // 12 GiB
    const size_t n = bytes / 16;
    u32x4 *d; uint32_t *out;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(d, 0x5A, bytes));
    for (int blocks : {2048, 8192, 65536}) {
        if (run<false, 4>("default loads, 4 x 16 B in flight", d, n, out, blocks)) return 1;
        if (run<true, 4>("nt loads, 4 x 16 B in flight", d, n, out, blocks)) return 1;
        if (run<true, 8>("nt loads, 8 x 16 B in flight", d, n, out, blocks)) return 1;
    }
    return 0;
}
