// Microbenchmark: v_mfma_i32_32x32x32_i8 issue rate alone, the bit->byte expansion alone, and both
// interleaved, at 1 and 2 waves per SIMD.  Cycles from s_memtime (shader clock), clock from
// s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2000;

__device__ __forceinline__ uint32_t spread0(uint32_t v, uint32_t k) { uint32_t r; asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(v), "v"(k)); return r; }
__device__ __forceinline__ uint32_t spread1(uint32_t v, uint32_t k) { uint32_t r; asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(v), "v"(k)); return r; }
__device__ __forceinline__ i32x4 expand16(uint32_t xs, uint32_t kmul) {
    const uint32_t y = xs & 0x0F0Fu, z = (xs >> 4) & 0x0F0Fu;
    i32x4 r;
    r.x = (int)(spread0(y, kmul) & 0x01010101u); r.y = (int)(spread0(z, kmul) & 0x01010101u);
    r.z = (int)(spread1(y, kmul) & 0x01010101u); r.w = (int)(spread1(z, kmul) & 0x01010101u);
    return r;
}

// MODE 0: 8 MFMA per iteration (2x4 accumulators), operands fixed.  MODE 1: 6 expansions per iteration (VALU only).
// MODE 2: 6 expansions + 8 MFMA per iteration (the Gram inner step).  MODE 3: 16 MFMA (4x4), MODE 4: 8 exp + 16 MFMA.
template <int MODE>
__global__ __launch_bounds__(64, (MODE == 3 || MODE == 4) ? 1 : 2) void k(const uint32_t *in, int *out, unsigned long long *stamps) {
    const uint32_t lane = threadIdx.x;
    constexpr int NA = (MODE == 3 || MODE == 4) ? 4 : 2;
    i32x16 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    uint32_t x[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) x[g] = in[lane + 64 * g];
    i32x4 fa[4], fb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { fa[g] = expand16(x[g], 0x204081u); fb[g] = expand16(x[4 + g], 0x204081u); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int g = 0; g < NA; ++g) fa[g] = expand16(x[g] + it, 0x204081u);
#pragma unroll
            for (int g = 0; g < 4; ++g) fb[g] = expand16(x[4 + g] ^ it, 0x204081u);
        }
        if (MODE != 1) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int g = 0; g < NA; ++g) acc[0][0][g] += fa[g].x ^ fa[g].y ^ fa[g].z ^ fa[g].w;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[1][0][g] += fb[g].x ^ fb[g].y ^ fb[g].z ^ fb[g].w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) s ^= acc[a][b][e];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
int run(const char *name, int waves_per_simd) {
    const int grid = 256 * 4 * waves_per_simd;
    uint32_t *din; int *dout; unsigned long long *dst;
    CHECK(hipMalloc(&din, 512 * 4)); CHECK(hipMalloc(&dout, (size_t)grid * 64 * 4)); CHECK(hipMalloc(&dst, (size_t)grid * 16));
    uint32_t h[512]; for (int i = 0; i < 512; ++i) h[i] = 0x9E3779B9u * (i + 1);
    CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, din, dout, dst);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, din, dout, dst);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long *hs = (unsigned long long *)malloc((size_t)grid * 16);
    CHECK(hipMemcpy(hs, dst, (size_t)grid * 16, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < grid; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    printf("%-28s waves/SIMD=%d  wall %.3f ms  %.0f shader cycles per iteration per wave, clock %.2f GHz\n", name, waves_per_simd, ms,
           cyc / ITER, cyc / (rt * 10.0) / 1e0 * 1e-3 * 1e3 / 1e3 * 1.0);
    free(hs); hipFree(din); hipFree(dout); hipFree(dst);
    return 0;
}
int main() {
    for (int w : {1, 2}) {
        if (run<0>("8 MFMA i8 32x32x32", w)) return 1;
        if (run<1>("6 expand16", w)) return 1;
        if (run<2>("6 expand16 + 8 MFMA", w)) return 1;
    }
    if (run<3>("16 MFMA (4x4 acc)", 1)) return 1;
    if (run<4>("8 expand16 + 16 MFMA", 1)) return 1;
    return 0;
}
