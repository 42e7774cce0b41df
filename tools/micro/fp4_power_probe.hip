// Probe: what MAC rate does the chip SUSTAIN on FP4 (E2M1) MFMAs, and does it depend on the operand data?
// The all-pairs Gram kernel runs at MfmaUtil 88 % of a clock that drops to ~1.6 GHz; this probe
// separates the clock question from the kernel: bare MFMA loops over the whole chip for tens of milliseconds with
//   * operand density 0 / 5 % / 50 % / 100 % of the 0b0010 (= 1.0) nibbles the Gram kernel feeds,
//   * v_mfma_f32_32x32x64_f8f6f4 (3 x 3 accumulator blocks, the Gram tile) and v_mfma_f32_16x16x128_f8f6f4 (6 x 6),
//   * 1 and 2 waves per SIMD.
// Output: one line per case with the achieved PMAC/s over the launch (HIP events) and the s_memtime cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(const uint32_t *in, float *out, int iters, unsigned long long *stamps) {
    const uint32_t lane = threadIdx.x & 63;
    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    i32x8 fa[3], fb[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        fa[g] = (i32x8){0, 0, 0, 0, 0, 0, 0, 0};
        fb[g] = (i32x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[g][q] = (int)in[(threadIdx.x + 256 * (g * 4 + q)) & 8191];
            fb[g][q] = (int)in[(threadIdx.x + 256 * (12 + g * 4 + q)) & 8191];
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[a][b][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
    (void)lane;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(const uint32_t *in, float *out, int iters, unsigned long long *stamps) {
    f32x4 acc[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;
    i32x8 fa[6], fb[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        fa[g] = (i32x8){0, 0, 0, 0, 0, 0, 0, 0};
        fb[g] = (i32x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[g][q] = (int)in[(threadIdx.x + 256 * (g * 4 + q)) & 8191];
            fb[g][q] = (int)in[(threadIdx.x + 256 * (24 + g * 4 + q) + 77) & 8191];
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[a][b][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
}

int main() {
    const int N = 8192;
    uint32_t *h = (uint32_t *)malloc(N * 4), *din;
    float *dout;
    unsigned long long *dst, st;
    CHECK(hipMalloc(&din, N * 4));
    CHECK(hipMalloc(&dout, 4 << 20));
    CHECK(hipMalloc(&dst, 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double dens[4] = {0.0, 0.05, 0.5, 1.0};
    for (int rep = 0; rep < 2; ++rep)
        for (int d = 0; d < 4; ++d) {
            srand(11 + d);
            for (int i = 0; i < N; ++i) {
                uint32_t v = 0;
                for (int b = 0; b < 8; ++b)
                    if ((double)rand() / RAND_MAX < dens[d]) v |= 2u << (4 * b);  // nibble 0b0010 = 1.0
                h[i] = v;
            }
            CHECK(hipMemcpy(din, h, N * 4, hipMemcpyHostToDevice));
            for (int kind = 0; kind < 4; ++kind) {
                // kind 0: 32x32x64, 2 waves/SIMD; 1: 32x32x64, 1 wave/SIMD; 2: 16x16x128, 2 waves/SIMD; 3: 16x16x128 1 wave/SIMD
                const int iters = kind < 2 ? 60000 : 15000;  // 9 x 32 cycles vs 36 x 16 cycles per iteration; ~25-50 ms
                const int it2 = (kind & 1) ? iters * 2 : iters;
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k32<4>, dim3(512), dim3(256), 0, 0, din, dout, it2, dst);
                if (kind == 1) hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, din, dout, it2, dst);
                if (kind == 2) hipLaunchKernelGGL(k16<4>, dim3(512), dim3(256), 0, 0, din, dout, it2, dst);
                if (kind == 3) hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, din, dout, it2, dst);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                CHECK(hipMemcpy(&st, dst, 8, hipMemcpyDeviceToHost));
                const double waves = ((kind & 1) ? 256.0 : 512.0) * 4;
                const double mf_per_iter = kind < 2 ? 9.0 : 36.0;
                const double macs = waves * it2 * mf_per_iter * (kind < 2 ? 32.0 * 32 * 64 : 16.0 * 16 * 128);
                printf("rep %d density %.2f %s %d wave(s)/SIMD: %.2f ms, %.3f PMAC/s (%.1f %% of 5.03), %.1f memtime ticks per MFMA per wave\n",
                       rep, dens[d], kind < 2 ? "32x32x64 " : "16x16x128", (kind & 1) ? 1 : 2, ms, macs / ms / 1e12,
                       macs / ms / 1e12 / 5.03 * 100, (double)st / (it2 * mf_per_iter));
                fflush(stdout);
            }
        }
    return 0;
}
