// Microbenchmark: issue rate of v_and_b32, v_bcnt_u32_b32, v_mul_u32_u24, v_perm_b32 on gfx950.
// Each kernel runs ITER iterations of 32 independent instructions per lane; grid fills every SIMD
// with `waves` waves.  Prints lane-ops/s and cycles per wave-instruction per SIMD (at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = seed + threadIdx.x * 33 + i;
    uint32_t m = seed | 0x01010101u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 3) asm volatile("v_perm_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 4) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_bfe_u32 %0, %0, 3, 4" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (OP == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
int run(const char *name, int blocks_per_cu) {
    uint32_t *d;
    const int grid = 256 * blocks_per_cu;
    CHECK(hipMalloc(&d, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)grid * 4 * ITER * 32;      // wave-instructions
    const double per_simd = wave_instr / (256.0 * 4);
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.2e lane-ops/s\n", name, blocks_per_cu, ms, cyc,
           wave_instr * 64 / (ms * 1e-3));
    CHECK(hipFree(d));
    return 0;
}
int main() {
    for (int w : {1, 2, 4}) {
        if (run<0>("v_and_b32", w)) return 1;
        if (run<1>("v_bcnt_u32_b32", w)) return 1;
        if (run<2>("v_mul_u32_u24", w)) return 1;
        if (run<3>("v_perm_b32", w)) return 1;
        if (run<4>("v_mul_lo_u32", w)) return 1;
        if (run<5>("v_bfe_u32", w)) return 1;
        if (run<6>("v_xor_b32", w)) return 1;
        if (run<7>("v_add_u32", w)) return 1;
    }
    return 0;
}
