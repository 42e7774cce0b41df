// ehh.hip — extended haplotype homozygosity on the resident bit matrix: the reference's
// calc_EHH (scripts/wip/ehhgfa.py:6-21, ehh2.py:76-89).
//
// EHH[i] = round(#{pairs j<k identical on the first i+1 sites} / (m(m-1)/2), 3).  A pair stays
// "homozygous" exactly until its first differing site f_jk, so the whole vector is the suffix count
// of the histogram of first-mismatch positions:
//     pairs(i) = C(m,2) - #{f_jk <= i}
//   1. ehh_transpose_kernel   SB64 window -> word-major [64-site block][haplotype] u64 (ballot
//                             transpose, window edges masked off), so that lane = haplotype reads
//                             are coalesced
//   2. ehh_first_diff_kernel  one thread per pair: XOR the two rows block by block until the first
//                             non-zero word; integer atomic into hist[f]
//   3. ehh_finalize_kernel    one workgroup: scan of hist, then the reference's division and
//                             CPython round(, 3) in fp64
#include <vector>

#include "device_utils.h"
#include "internal.h"

namespace impop {

__global__ __launch_bounds__(256) void ehh_transpose_kernel(const uint32_t *__restrict__ sb, uint32_t wps, uint32_t G, uint32_t r,
                                                            uint64_t site_begin, uint64_t site_end, uint32_t n_pad,
                                                            uint64_t *__restrict__ wm) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t blk0 = site_begin >> 6;
    const uint64_t bi = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t b = blk0 + bi;
    if (b * 64 >= site_end) return;  // wave-uniform
    uint64_t edge = ~0ull;           // sites of this block inside [site_begin, site_end)
    if (b * 64 < site_begin) edge &= ~0ull << (site_begin - b * 64);
    if (site_end - b * 64 < 64) edge &= (1ull << (site_end - b * 64)) - 1ull;
    for (uint32_t k = 0; k < wps; ++k) {
        const uint32_t w = sb[sb_index(wps, G, r, b, lane, k)];
        uint64_t keep = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t m = __ballot((w >> j) & 1u);
            if (lane == (uint32_t)j) keep = m;
        }
        if (lane < 32) wm[bi * n_pad + 32 * k + lane] = keep & edge;
    }
}

// grid = (ceil(m/256), m): blockIdx.y = position a in the member list, threads = positions b > a
__global__ __launch_bounds__(256) void ehh_first_diff_kernel(const uint64_t *__restrict__ wm, uint32_t n_pad, uint64_t n_blk,
                                                             const uint32_t *__restrict__ idx, uint32_t m, uint64_t first_site_off,
                                                             uint64_t W, int reverse, unsigned long long *__restrict__ hist) {
    const uint32_t a = blockIdx.y;
    const uint32_t bpos = blockIdx.x * 256 + threadIdx.x;
    if (bpos <= a || bpos >= m) return;
    const uint32_t ra = idx[a], rb = idx[bpos];
    // bit position p of block t is window site t*64 + p - first_site_off
    if (!reverse) {
        for (uint64_t t = 0; t < n_blk; ++t) {
            const uint64_t x = wm[t * n_pad + ra] ^ wm[t * n_pad + rb];
            if (x) {
                const uint64_t f = t * 64 + (uint64_t)__builtin_ctzll(x) - first_site_off;
                atomicAdd(&hist[f], 1ull);
                return;
            }
        }
    } else {
        for (uint64_t t = n_blk; t-- > 0;) {
            const uint64_t x = wm[t * n_pad + ra] ^ wm[t * n_pad + rb];
            if (x) {
                const uint64_t pos = t * 64 + (63 - (uint64_t)__builtin_clzll(x)) - first_site_off;
                atomicAdd(&hist[W - 1 - pos], 1ull);
                return;
            }
        }
    }
}

constexpr int EHH_FT = 1024;
__global__ __launch_bounds__(EHH_FT) void ehh_finalize_kernel(const unsigned long long *__restrict__ hist, uint64_t W, uint32_t m,
                                                              double *__restrict__ out) {
    __shared__ unsigned long long part[EHH_FT];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (W + EHH_FT - 1) / EHH_FT;
    const uint64_t lo = (uint64_t)tid * per < W ? (uint64_t)tid * per : W;
    const uint64_t hi = lo + per < W ? lo + per : W;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += hist[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < EHH_FT; off <<= 1) {  // inclusive Hillis-Steele scan over the chunk sums
        const unsigned long long v = tid >= (uint32_t)off ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = (unsigned long long)m * (m - 1) / 2;
    const double denom = (double)((unsigned long long)m * (m - 1)) / 2.0;  // ehhgfa.py:20: m*(m-1)/2 (true division)
    unsigned long long broken = part[tid] - s;                            // pairs whose first difference is before lo
    for (uint64_t i = lo; i < hi; ++i) {
        broken += hist[i];
        out[i] = py_round((double)(total - broken) / denom, 3);
    }
}

__global__ void ehh_fill_kernel(double *out, uint64_t W, double v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) out[i] = v;
}

}  // namespace impop

using namespace impop;

IMPOP_API int impop_ehh(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end, const uint64_t *mask,
                        int reverse, double *ehh_out_host, uint32_t *n_members) {
    REQUIRE(ctx && m, "impop_ehh: NULL argument");
    NOT_COMPACT(m, "impop_ehh");
    REQUIRE(site_begin <= site_end && site_end <= m->g.n_site, "impop_ehh: bad site range");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = m->g.n_hap;
    std::vector<uint32_t> idx;
    for (uint32_t i = 0; i < n; ++i)
        if (!mask || ((mask[i >> 6] >> (i & 63)) & 1ull)) idx.push_back(i);
    const uint32_t mm = (uint32_t)idx.size();
    if (n_members) *n_members = mm;
    const uint64_t W = site_end - site_begin;
    if (!W) return IMPOP_OK;  // calc_EHH returns an empty vector
    REQUIRE(ehh_out_host, "impop_ehh: out is NULL");
    REQUIRE(mm <= 65535, "impop_ehh: more than 65535 member haplotypes not supported");
    const uint64_t blk0 = site_begin >> 6, n_blk = ((site_end + 63) >> 6) - blk0;
    const uint32_t n_pad = m->g.wps * 32;
    const size_t wm_bytes = (size_t)n_blk * n_pad * 8;
    REQUIRE(wm_bytes <= (64ull << 30), "impop_ehh: window too large (%llu MiB of transposed scratch)",
            (unsigned long long)(wm_bytes >> 20));
    const size_t o_hist = (wm_bytes + 255) / 256 * 256, o_out = o_hist + (W * 8 + 255) / 256 * 256,
                 o_idx = o_out + (W * 8 + 255) / 256 * 256;
    void *d = nullptr;
    int rc = ctx_scratch(ctx, o_idx + (size_t)(mm ? mm : 1) * 4, &d);
    if (rc) return rc;
    uint64_t *d_wm = (uint64_t *)d;
    unsigned long long *d_hist = (unsigned long long *)((char *)d + o_hist);
    double *d_out = (double *)((char *)d + o_out);
    uint32_t *d_idx = (uint32_t *)((char *)d + o_idx);
    if (mm < 2) {  // ehhgfa.py:17-18: fewer than two haplotypes -> every entry 500
        hipLaunchKernelGGL(ehh_fill_kernel, dim3((uint32_t)((W + 255) / 256)), dim3(256), 0, ctx->stream, d_out, W, 500.0);
    } else {
        REQUIRE((n_blk + 3) / 4 < 0x7FFFFFFFull, "impop_ehh: range too long");
        HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), (size_t)mm * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemsetAsync(d_hist, 0, W * 8, ctx->stream));
        hipLaunchKernelGGL(ehh_transpose_kernel, dim3((uint32_t)((n_blk + 3) / 4)), dim3(256), 0, ctx->stream, m->d_sb, m->g.wps,
                           m->g.G, m->g.r, site_begin, site_end, n_pad, d_wm);
        hipLaunchKernelGGL(ehh_first_diff_kernel, dim3((mm + 255) / 256, mm), dim3(256), 0, ctx->stream, d_wm, n_pad, n_blk, d_idx,
                           mm, site_begin - blk0 * 64, W, reverse ? 1 : 0, d_hist);
        hipLaunchKernelGGL(ehh_finalize_kernel, dim3(1), dim3(EHH_FT), 0, ctx->stream, d_hist, W, mm, d_out);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ehh_out_host, d_out, W * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}
