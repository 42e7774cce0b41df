// Microbenchmark for a candidate "SB64T" layout: per 64-site block 14 full dwords per site in the usual
// wave-interleaved granules (448 haplotypes) + the last 17 haplotypes as 64-bit hap-major words (bit l = site l),
// 3728 B per block (16-byte aligned) instead of 3840 B for 15 dwords per site.  Emulates the scan's per-site work
// (count, count & A, count & B, S tests, two products) on both layouts over the same number of sites and prints the
// time per pass: is the 2.9 % byte saving still a time saving once the tail costs scalar loads and per-lane bit adds?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct Masks { uint32_t a[15], b[15]; };

__device__ __forceinline__ void addc(uint32_t &c, uint64_t mask) {  // c += bit `lane` of mask (one VALU instruction)
    uint64_t dummy;
    asm volatile("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(c), "=s"(dummy) : "s"(mask));
}

// LAYOUT 0: 15 dwords per site (3 x dwordx4 + 1 x dwordx3), 3840 B per block.  LAYOUT 1: 14 dwords (3 x dwordx4 + dwordx2)
// + 17 tail words, 3728 B per block.
template <int LAYOUT, int TILE>
__global__ __launch_bounds__(256, 6) void scan(const uint32_t *__restrict__ sb, uint64_t n_block, Masks mk, uint32_t nA, uint32_t nB,
                                               uint64_t tailA, uint64_t tailB /* which tail haplotypes are in A / B (bit j) */,
                                               unsigned long long *__restrict__ out) {
    constexpr uint32_t BLK = LAYOUT ? 932u : 960u;  // dwords per block
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t s_all = 0, qa = 0, qb = 0, qab = 0;
    const uint64_t b0 = (uint64_t)blockIdx.x * TILE, b1 = b0 + TILE < n_block ? b0 + TILE : n_block;
    auto one_block = [&](uint64_t b) {
        const uint32_t *blk = sb + b * BLK;
        uint32_t w[15];
        u32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(blk + lane * 4));
        u32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(blk + 256 + lane * 4));
        u32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(blk + 512 + lane * 4));
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        w[8] = v2.x; w[9] = v2.y; w[10] = v2.z; w[11] = v2.w;
        uint32_t c = 0, cA = 0, cB = 0;
        if (LAYOUT == 0) {
            const uint32_t *l = blk + 768 + lane * 3;
            w[12] = __builtin_nontemporal_load(l); w[13] = __builtin_nontemporal_load(l + 1); w[14] = __builtin_nontemporal_load(l + 2);
#pragma unroll
            for (int k = 0; k < 15; ++k) { c += __popc(w[k]); cA += __popc(w[k] & mk.a[k]); cB += __popc(w[k] & mk.b[k]); }
        } else {
            const u32x2 v3 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(blk + 768 + lane * 2));
            w[12] = v3.x; w[13] = v3.y;
            const uint64_t *tail = reinterpret_cast<const uint64_t *>(blk + 896);  // wave-uniform: scalar loads
#pragma unroll
            for (int k = 0; k < 14; ++k) { c += __popc(w[k]); cA += __popc(w[k] & mk.a[k]); cB += __popc(w[k] & mk.b[k]); }
#pragma unroll
            for (int j = 0; j < 17; ++j) {
                const uint64_t t = tail[j];
                addc(c, t);
                if ((tailA >> j) & 1) addc(cA, t);  // wave-uniform (scalar) branches
                if ((tailB >> j) & 1) addc(cB, t);
            }
        }
        s_all += (c - 1u) < 464u;
        qa += __umul24(cA, nA - cA); qb += __umul24(cB, nB - cB); qab += __umul24(cA, nB - cB) + __umul24(cB, nA - cA);
    };
    uint64_t b = b0 + wave;
    for (; b + 4 < b1; b += 8) { one_block(b); one_block(b + 4); }
    for (; b < b1; b += 4) one_block(b);
    const uint32_t t = s_all ^ qa ^ qb ^ qab;  // depends on every loaded word: the loads cannot be dropped
    if (t == 0x12345678u) out[0] = t;
}

template <int LAYOUT>
int run(const char *name, const uint32_t *d, uint64_t n_block, unsigned long long *out) {
    constexpr int TILE = 68;
    Masks mk;
    for (int k = 0; k < 15; ++k) { mk.a[k] = k < 4 ? 0xFFFFFFFFu : (k == 4 ? 0xFFFu : 0u); mk.b[k] = (k == 4 ? 0xFFFFF000u : (k > 4 && k < 7 ? 0xFFFFFFFFu : (k == 7 ? 0xFFFFu : 0u))); }
    const uint32_t grid = (uint32_t)((n_block + TILE - 1) / TILE);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 7; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((scan<LAYOUT, TILE>), dim3(grid), dim3(256), 0, 0, d, n_block, mk, 140u, 100u, 0ull, 0ull, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    const double bytes = (double)n_block * (LAYOUT ? 3728.0 : 3840.0), algo = (double)n_block * 64 * 465 / 8;
    printf("%-44s %.3f ms  layout %.0f GB/s  algorithmic %.0f GB/s (%.3f of 8 TB/s)\n", name, best, bytes / best / 1e6, algo / best / 1e6, algo / best / 8e9);
    // tail haplotypes inside the masks (worst case: all 17 in A, and in B)
    best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((scan<LAYOUT, TILE>), dim3(grid), dim3(256), 0, 0, d, n_block, mk, 140u, 100u, 0x1FFFFull, 0x1FFFFull, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    if (LAYOUT) printf("%-44s %.3f ms  (all 17 tail haplotypes in A and in B)\n", "", best);
    return 0;
}

int main() {
    const uint64_t n_block = 242700000ull / 64;  // chr2
    uint32_t *d; unsigned long long *out;
    CHECK(hipMalloc(&d, n_block * 3840 + 4096)); CHECK(hipMalloc(&out, 16));
    CHECK(hipMemset(d, 0x5A, n_block * 3840 + 4096)); CHECK(hipMemset(out, 0, 16));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>("SB64  15 dwords/site, 3840 B/block", d, n_block, out)) return 1;
        if (run<1>("SB64T 14 dwords + 17 tail words, 3728 B/block", d, n_block, out)) return 1;
    }
    return 0;
}
