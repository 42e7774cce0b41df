// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands built from RAW BIT PLANES.
//  * exactness: G[i][j] = sum over sites of a_i & b_j from four plane MFMAs per 4 raw dwords per lane
//      variant 0: every plane shifted to nibble value 0b0010 (= 1.0), scales = 127 (2^0)
//      variant 1: same operands, scale operands = 0 (does the compiler pick the unscaled form / what does 0 mean?)
//      variant 2: unshifted planes (0.5 / 1 / 2 / 2) with A-side scales 129 / 127 / 125 / 125
//  * rate: cycles per MFMA for a 3x3 accumulator block, 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__device__ __forceinline__ void planes(const uint32_t (&x)[4], int p, i32x8 &f) {
    f = (i32x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v;
        if (VARIANT == 2) v = p == 3 ? (x[q] >> 1) & 0x44444444u : x[q] & (0x11111111u << p);
        else v = p == 0 ? (x[q] << 1) & 0x22222222u : (x[q] >> (p - 1)) & 0x22222222u;
        f[q] = (int)v;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(64) void exact_kernel(const uint32_t *ra, const uint32_t *rb, float *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t xa[4], xb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { xa[q] = ra[lane * 4 + q]; xb[q] = rb[lane * 4 + q]; }  // [row=lane&31][half=lane>>5][q]
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        i32x8 fa, fb;
        planes<VARIANT>(xa, p, fa);
        planes<VARIANT>(xb, p, fb);
        if (VARIANT == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        else if (VARIANT == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 4, 4, 0, 0, 0, 0);
        else {
            const int sa = p == 0 ? 0x81818181 : p == 1 ? 0x7F7F7F7F : 0x7D7D7D7D;
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 4, 4, 0, sa, 0, 0x7F7F7F7F);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[row * 32 + col] = acc[e];
    }
}

constexpr int ITER = 2000;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rate_kernel(const uint32_t *in, float *out, unsigned long long *stamps) {
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
        for (int q = 0; q < 4; ++q) { fa[g][q] = (int)(in[lane + 64 * (g * 4 + q)] & 0x22222222u); fb[g][q] = (int)(in[lane + 64 * (12 + g * 4 + q)] & 0x22222222u); }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[a][b][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main() {
    uint32_t ha[256], hb[256];
    srand(7);
    for (int i = 0; i < 256; ++i) { ha[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); hb[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); }
    uint32_t *da, *db, *din; float *dout; unsigned long long *dst;
    CHECK(hipMalloc(&da, sizeof ha)); CHECK(hipMalloc(&db, sizeof hb)); CHECK(hipMalloc(&dout, 1 << 20)); CHECK(hipMalloc(&dst, 64));
    CHECK(hipMalloc(&din, 64 * 24 * 4));
    CHECK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    CHECK(hipMemset(din, 0x22, 64 * 24 * 4));
    float want[32 * 32];
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int h = 0; h < 2; ++h)
                for (int q = 0; q < 4; ++q) s += __builtin_popcount(ha[(i + 32 * h) * 4 + q] & hb[(j + 32 * h) * 4 + q]);
            want[i * 32 + j] = (float)s;
        }
    float got[32 * 32];
    for (int v = 0; v < 3; ++v) {
        if (v == 0) hipLaunchKernelGGL(exact_kernel<0>, dim3(1), dim3(64), 0, 0, da, db, dout);
        if (v == 1) hipLaunchKernelGGL(exact_kernel<1>, dim3(1), dim3(64), 0, 0, da, db, dout);
        if (v == 2) hipLaunchKernelGGL(exact_kernel<2>, dim3(1), dim3(64), 0, 0, da, db, dout);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
        int bad = 0; double ratio = 0;
        for (int i = 0; i < 1024; ++i) { if (got[i] != want[i]) ++bad; if (want[i] > 0) ratio = got[i] / want[i]; }
        printf("variant %d: %d / 1024 mismatches (got[0]=%g want[0]=%g, last ratio %g)\n", v, bad, got[0], want[0], ratio);
    }
    unsigned long long st[2];
    hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * 4), dim3(64), 0, 0, din, dout, dst);
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    printf("rate 1 wave/SIMD : %.2f memtime ticks per MFMA, %.2f ns per MFMA (memrealtime @100 MHz)\n", (double)st[0] / (ITER * 9.0), (double)st[1] * 10.0 / (ITER * 9.0));
    hipLaunchKernelGGL(rate_kernel<4>, dim3(256 * 2), dim3(256), 0, 0, din, dout, dst);
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    printf("rate 2 waves/SIMD: %.2f memtime ticks per MFMA per wave, %.2f ns per MFMA per wave\n", (double)st[0] / (ITER * 9.0), (double)st[1] * 10.0 / (ITER * 9.0));
    // wall-clock rate over the whole chip
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<4>, dim3(256 * 2), dim3(256), 0, 0, din, dout, dst);
    hipEventRecord(e1); CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = 256.0 * 2 * 4 * ITER * 9 * 32 * 32 * 64;
    printf("chip: %.3f ms, %.2f PMAC/s (= %.2f PFLOP/s fp4)\n", ms, macs / ms / 1e12, 2 * macs / ms / 1e12);
    return 0;
}
