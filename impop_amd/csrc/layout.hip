// layout.hip — presence-matrix residency: upload, SB64 <-> hap-major transposition on the
// device (wave64 readlane / ballot bit transposes), the synthetic generator and download.
#include <string.h>

#include <algorithm>

#include <vector>

#include <algorithm>

#include "device_utils.h"
#include "internal.h"

namespace impop {

static SbGeom make_geom(uint32_t n_hap, uint64_t n_site) {
    SbGeom g;
    g.n_hap = n_hap;
    g.wps = (n_hap + 31) / 32;
    g.G = (g.wps + 3) / 4;
    g.r = g.wps - 4 * (g.G - 1);
    g.n_site = n_site;
    g.n_block = (n_site + 63) / 64;
    return g;
}

// ---- hap-major -> SB64 --------------------------------------------------------------
// One wave per 64-site block.  For each 32-haplotype dword k, lanes 0..31 fetch the
// 64-site word of haplotype 32k+lane; v_readlane broadcasts each of them and every lane
// (= site) picks its own bit: a 32x64 bit transpose in 32 readlane+bfe+lshl_or steps.
__global__ __launch_bounds__(256) void hm_to_sb_kernel(const uint32_t *__restrict__ hm, uint64_t hm_stride,
                                                       uint32_t n_rows, uint32_t wps, uint32_t G, uint32_t r,
                                                       uint64_t n_block, uint32_t *__restrict__ sb) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_block) return;  // wave-uniform
    for (uint32_t k = 0; k < wps; ++k) {
        uint64_t w = 0;
        const uint32_t row = 32 * k + lane;
        if (lane < 32 && row < n_rows) {
            const uint32_t *p = hm + (uint64_t)row * hm_stride + 2 * b;
            w = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
        }
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint64_t wj = __shfl(w, j, 64);
            out |= (uint32_t)((wj >> lane) & 1ull) << j;
        }
        sb[sb_index(wps, G, r, b, lane, k)] = out;
    }
}

// ---- SB64 -> hap-major ---------------------------------------------------------------
// One wave per block; lane = site.  The 64-site hap-major word of haplotype 32k+j is bit j of every site's dword k,
// gathered by transposing the two 32 x 32 bit matrices a dword k forms over the wave's halves (below).
// phi_row != 0xFFFFFFFF (RB32 operand of the all-pairs path only): MINOR-ALLELE POLARITY.  A site more than half of the n_hap
// haplotypes carry is stored complemented, and the set of complemented sites is stored as one more row — row phi_row = n_hap, in
// the padding up to the next multiple of 96 — so that the Gram kernel delivers, for free, what undoes it:
//     I_ij = I'_ij + I'_pp - I'_ip - I'_jp      (p = phi_row; primes = the stored polarity; gram_unflip_kernel, pairwise.hip)
// while the Hamming distance a_i + a_j - 2 I_ij — all the `match` identity needs — is the same in either polarity.  Why: the FP4
// matrix cores are power-limited and their power depends on the operands (tools/micro/fp4_power_probe.hip: 4.0-4.5 PMAC/s sustained
// at 50 % ones, 4.8-4.9 at 5 % or 0 %); a presence matrix is 50 % ones only by the accident of which allele was called 1.  Measured on
// the bench matrix (ancestral polarity random): Gram 8.0 -> 7.0 ms per 4096 windows of 465 x 50 kb.
__global__ __launch_bounds__(256) void sb_to_hm_kernel(const uint32_t *__restrict__ sb, uint32_t wps, uint32_t G,
                                                       uint32_t r, uint64_t blk_begin, uint64_t blk_end,
                                                       uint32_t *__restrict__ hm, uint64_t hm_stride,
                                                       uint32_t n_rows, uint64_t rb_nb, uint32_t phi_row, uint32_t n_hap) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = blk_begin + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= blk_end) return;  // wave-uniform
    uint32_t flip = 0;  // all ones: this lane's site is stored complemented
    if (phi_row != 0xFFFFFFFFu) {
        uint32_t c = 0;
        for (uint32_t k = 0; k < wps; ++k) c += (uint32_t)__popc(sb[sb_index(wps, G, r, b, lane, k)]);
        flip = 2 * c > n_hap ? 0xFFFFFFFFu : 0u;
    }
    // Dword k of site `lane` holds haplotypes 32k .. 32k+31: each half of the wave is a 32 x 32 bit matrix (site x haplotype)
    // to transpose.  Five rounds of swapping the off-diagonal s x s blocks of every 2s x 2s block with lane ^ s (s = 16 .. 1):
    // one cross-lane move + four bit operations per round, against 32 ballots per dword before (18.8 ms per 12 GB matrix;
    // 10.8 ms with the butterfly and whole-granule loads).  Afterwards lane j of half h holds, for haplotype 32k + j, the
    // sites 32h .. 32h+31 of the block: dword h of that row's pair.
    const uint32_t half = lane >> 5, j = lane & 31;
    const uint32_t *blk = sb + b * 64ull * wps;
    uint4 q = {0u, 0u, 0u, 0u};  // the granule dword k comes out of
    for (uint32_t k = 0; k < wps; ++k) {
        // whole granules as one 16-byte load per lane (dword k alone is a 4-byte read at a 16-byte stride: a quarter of
        // every cache line per instruction); the last, shorter granule dword by dword
        uint32_t w;
        if ((k >> 2) + 1 < G || r == 4) {
            static_assert(sizeof(uint4) == 16, "granule");
            if ((k & 3) == 0) q = *reinterpret_cast<const uint4 *>(blk + (uint64_t)(k >> 2) * 256 + lane * 4);
            w = (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
        } else {
            w = sb[sb_index(wps, G, r, b, lane, k)];
        }
        if (flip) w ^= 32 * k + 32 <= n_hap ? 0xFFFFFFFFu : 32 * k < n_hap ? (1u << (n_hap - 32 * k)) - 1u : 0u;  // real haplotypes only
#pragma unroll
        for (uint32_t s = 16; s != 0; s >>= 1) {
            const uint32_t M = s == 16 ? 0x0000FFFFu : s == 8 ? 0x00FF00FFu : s == 4 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
            const uint32_t o = (uint32_t)__shfl_xor((int)w, (int)s, 64);
            w = (lane & s) ? ((w & ~M) | ((o >> s) & M)) : ((w & M) | ((o & M) << s));
        }
        const uint32_t row = 32 * k + j;
        if (row < n_rows && row != phi_row) {
            const uint64_t d = 2 * (b - blk_begin);  // first of the two dwords of this 64-site block
            uint32_t *p = rb_nb ? hm + ((((uint64_t)(row >> 5) * rb_nb + (d >> 1)) * 32 + (row & 31)) * 2)
                                : hm + (uint64_t)row * hm_stride + d;
            p[half] = w;
        }
    }
    if (phi_row != 0xFFFFFFFFu) {  // the complemented sites of this block as the 64-site word of row phi_row
        const uint64_t bal = __ballot(flip != 0);
        if (j == 0) {
            const uint64_t d = 2 * (b - blk_begin);
            uint32_t *p = hm + ((((uint64_t)(phi_row >> 5) * rb_nb + (d >> 1)) * 32 + (phi_row & 31)) * 2);
            p[half] = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        }
    }
}

// zero the bits of sites >= n_site in the last used dword pair of every hap-major row
__global__ void hm_clear_tail_kernel(uint32_t *hm, uint64_t hm_stride, uint32_t n_rows, uint64_t n_site,
                                     uint64_t n_dwords_used) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint32_t *p = hm + (uint64_t)row * hm_stride;
    const uint64_t first = n_site >> 5;
    if (first < n_dwords_used) {
        const uint32_t keepbits = (uint32_t)(n_site & 31);
        p[first] &= keepbits ? ((1u << keepbits) - 1u) : 0u;
        for (uint64_t d = first + 1; d < n_dwords_used; ++d) p[d] = 0;
    }
}

int launch_hm_to_sb(impop_ctx *ctx, const uint32_t *d_hm, uint64_t hm_stride, const SbGeom &g, uint32_t *d_sb) {
    if (g.n_block == 0) return IMPOP_OK;
    const uint64_t grid = (g.n_block + 3) / 4;
    REQUIRE(grid < 0x7FFFFFFFull, "matrix too long for one launch (%llu blocks)", (unsigned long long)g.n_block);
    const uint32_t n_rows = (g.n_hap + 95) / 96 * 96;
    hipLaunchKernelGGL(hm_to_sb_kernel, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, d_hm, hm_stride, n_rows, g.wps,
                       g.G, g.r, g.n_block, d_sb);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

int launch_sb_to_hm(impop_ctx *ctx, const uint32_t *d_sb, const SbGeom &g, uint64_t blk_begin, uint64_t blk_end,
                    uint32_t *d_hm, uint64_t hm_stride, uint32_t n_rows, uint64_t rb_nb, uint32_t phi_row) {
    if (blk_end <= blk_begin) return IMPOP_OK;
    REQUIRE(phi_row == 0xFFFFFFFFu || (rb_nb != 0 && phi_row == g.n_hap && phi_row < n_rows), "launch_sb_to_hm: bad phi_row");
    const uint64_t grid = (blk_end - blk_begin + 3) / 4;
    REQUIRE(grid < 0x7FFFFFFFull, "range too long for one launch");
    hipLaunchKernelGGL(sb_to_hm_kernel, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, d_sb, g.wps, g.G, g.r, blk_begin,
                       blk_end, d_hm, hm_stride, n_rows, rb_nb, phi_row, g.n_hap);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

// ---- synthetic generator ---------------------------------------------------------------
struct SynthDev {
    uint64_t seed;
    uint32_t n_founder;
    uint32_t thr_founder;  // p_founder * 2^32
    uint32_t thr_private;  // p_private_word * 2^32
};

// tables: fmask[f*wps + k] (haplotypes of founder f in dword k), then valid[k]
__global__ __launch_bounds__(256) void synth_sb_kernel(SynthDev p, const uint32_t *__restrict__ tables, uint32_t wps,
                                                       uint32_t G, uint32_t r, uint64_t n_block, uint64_t n_site,
                                                       uint64_t site0, uint32_t *__restrict__ sb) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_block) return;
    const bool live = b * 64 + lane < n_site;
    const uint64_t s = site0 + b * 64 + lane;  // the generator is counter-based on the GLOBAL site index: a slab is a cut of the whole
#ifdef IMPOP_SYNTH_ANC0  // experiment builds only (tools/): every ancestral allele 0 => a sparse matrix with the same pairwise distances
    const uint32_t anc = 0u;
#else
    const uint32_t anc = (synth_hash(p.seed, 1, s) & 1ull) ? 0xFFFFFFFFu : 0u;
#endif
    uint32_t fl = 0;
    for (uint32_t f = 0; f < p.n_founder; ++f)
        if ((uint32_t)(synth_hash(p.seed, 2 + f, s) >> 32) < p.thr_founder) fl |= 1u << f;
    const uint32_t *valid = tables + (uint64_t)p.n_founder * wps;
    for (uint32_t k = 0; k < wps; ++k) {
        uint32_t w = anc;
        uint32_t m = fl;
        while (m) {
            const int f = __ffs(m) - 1;
            m &= m - 1;
            w ^= tables[(uint64_t)f * wps + k];
        }
        const uint64_t h = synth_hash(p.seed, 64 + k, s);
        if ((uint32_t)(h >> 32) < p.thr_private) w ^= 1u << (uint32_t)(h & 31);
        w &= valid[k];
        sb[sb_index(wps, G, r, b, lane, k)] = live ? w : 0u;
    }
}

static int alloc_matrix(impop_ctx *ctx, uint32_t n_hap, uint64_t n_site, bool want_hm, impop_matrix **out) {
    impop_matrix *m = new impop_matrix();
    m->g = make_geom(n_hap, n_site);
    m->device = ctx->device;
    m->n_hap_pad = (n_hap + 95) / 96 * 96;  // Gram tiles are 96 haplotypes wide (3 row groups of 32)
    {   // minor-allele polarity of the all-pairs operand (sb_to_hm_kernel) needs one padding row for the set of complemented sites
        static const bool off = [] { const char *e = getenv("IMPOP_NO_POLARITY"); return e && e[0] == '1'; }();
        m->phi_row = (want_hm && !off && n_hap % 96 != 0 && n_hap <= 16384 /* gram_unflip_kernel's LDS */) ? n_hap : 0xFFFFFFFFu;
    }
    m->sb_bytes = m->g.n_block * 64ull * m->g.wps * 4ull;
    // one extra block of slack so software-pipelined kernels may prefetch one block past the end
    hipError_t e = hipMalloc((void **)&m->d_sb, m->sb_bytes + 64ull * m->g.wps * 4ull + 256);
    if (e != hipSuccess) {
        delete m;
        return hip_fail(e, "hipMalloc(SB64 matrix)", __FILE__, __LINE__);
    }
    e = hipMemsetAsync((char *)m->d_sb + m->sb_bytes, 0, 64ull * m->g.wps * 4ull + 256, ctx->stream);
    if (e != hipSuccess) {
        hipFree(m->d_sb);
        delete m;
        return hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
    }
    if (want_hm) {
        m->rb_nb = m->g.n_block + 8;  // 64-site cells per row group + slack (a 4-cell Gram step may overhang the last cell)
        m->rb_bytes = (uint64_t)(m->n_hap_pad / 32) * m->rb_nb * 32ull * 8ull;
        e = hipMalloc((void **)&m->d_rb, m->rb_bytes);
        if (e != hipSuccess) {
            hipFree(m->d_sb);
            delete m;
            return hip_fail(e, "hipMalloc(row-group-blocked matrix)", __FILE__, __LINE__);
        }
        e = hipMemsetAsync(m->d_rb, 0, m->rb_bytes, ctx->stream);
        if (e != hipSuccess) {
            hipFree(m->d_sb);
            hipFree(m->d_rb);
            delete m;
            return hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
        }
    }
    *out = m;
    return IMPOP_OK;
}

}  // namespace impop

using namespace impop;

IMPOP_API int impop_matrix_upload(impop_ctx *ctx, const uint64_t *bits, uint32_t n_hap, uint64_t n_site,
                                  uint64_t row_stride_words, uint32_t keep_flags, impop_matrix **out) {
    REQUIRE(ctx && out, "impop_matrix_upload: ctx/out is NULL");
    *out = nullptr;
    REQUIRE(n_hap >= 1, "impop_matrix_upload: n_hap must be >= 1");
    REQUIRE(n_hap <= (1u << 20), "impop_matrix_upload: n_hap %u too large", n_hap);
    const uint64_t words = (n_site + 63) / 64;
    REQUIRE(n_site == 0 || bits, "impop_matrix_upload: bits is NULL");
    REQUIRE(row_stride_words >= words, "impop_matrix_upload: row_stride_words %llu < ceil(n_site/64) = %llu",
            (unsigned long long)row_stride_words, (unsigned long long)words);
    HIP_TRY(hipSetDevice(ctx->device));
    impop_matrix *m = nullptr;
    const bool want_rb = (keep_flags & IMPOP_KEEP_HAP_MAJOR) != 0;
    int rc = alloc_matrix(ctx, n_hap, n_site, want_rb, &m);
    if (rc) return rc;
    // transient plain hap-major staging copy of the caller's rows (freed before returning)
    const uint64_t hm_stride = std::max<uint64_t>((m->g.n_block * 2 + 3) / 4 * 4, 4);
    const uint64_t hm_bytes = (uint64_t)m->n_hap_pad * hm_stride * 4ull;
    uint32_t *d_hm = nullptr;
    auto fail = [&](int code) {
        if (d_hm) hipFree(d_hm);
        impop_matrix_free(ctx, m);
        return code;
    };
    hipError_t e = hipMalloc((void **)&d_hm, hm_bytes);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc(upload staging)", __FILE__, __LINE__));
    e = hipMemsetAsync(d_hm, 0, hm_bytes, ctx->stream);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
    if (words) {
        e = hipMemcpy2DAsync(d_hm, hm_stride * 4ull, bits, row_stride_words * 8ull, words * 8ull, n_hap,
                             hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpy2DAsync(upload)", __FILE__, __LINE__));
        hipLaunchKernelGGL(hm_clear_tail_kernel, dim3((m->n_hap_pad + 255) / 256), dim3(256), 0, ctx->stream, d_hm, hm_stride,
                           m->n_hap_pad, n_site, words * 2);
        if ((e = hipGetLastError()) != hipSuccess) return fail(hip_fail(e, "hm_clear_tail_kernel", __FILE__, __LINE__));
    }
    rc = launch_hm_to_sb(ctx, d_hm, hm_stride, m->g, m->d_sb);
    if (rc) return fail(rc);
    if (want_rb) {
        rc = launch_sb_to_hm(ctx, m->d_sb, m->g, 0, m->g.n_block, m->d_rb, 0, m->n_hap_pad, m->rb_nb, m->phi_row);
        if (rc) return fail(rc);
        rc = ensure_segmap(ctx, m);  // a matrix kept for the all-pairs path gets its site bitmap (S per window) right away
        if (rc) return fail(rc);
    }
    e = hipStreamSynchronize(ctx->stream);  // the caller may free `bits` on return
    if (e != hipSuccess) return fail(hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
    hipFree(d_hm);
    *out = m;
    return IMPOP_OK;
}

IMPOP_API int impop_matrix_synthetic(impop_ctx *ctx, uint32_t n_hap, uint64_t n_site, const impop_synth_params *p,
                                     uint32_t keep_flags, impop_matrix **out) {
    return impop_matrix_synthetic_slab(ctx, n_hap, 0, n_site, p, keep_flags, out);
}

IMPOP_API int impop_matrix_synthetic_slab(impop_ctx *ctx, uint32_t n_hap, uint64_t site_begin, uint64_t n_site,
                                          const impop_synth_params *p, uint32_t keep_flags, impop_matrix **out) {
    REQUIRE(ctx && out && p, "impop_matrix_synthetic: NULL argument");
    *out = nullptr;
    REQUIRE(site_begin + n_site >= site_begin, "impop_matrix_synthetic_slab: site range wraps");
    REQUIRE(n_hap >= 1 && n_hap <= (1u << 20), "impop_matrix_synthetic: bad n_hap %u", n_hap);
    REQUIRE(p->n_founder >= 1 && p->n_founder <= 32, "impop_matrix_synthetic: n_founder must be 1..32");
    REQUIRE(p->p_founder >= 0 && p->p_founder < 1 && p->p_private_word >= 0 && p->p_private_word < 1,
            "impop_matrix_synthetic: probabilities must be in [0,1)");
    HIP_TRY(hipSetDevice(ctx->device));
    impop_matrix *m = nullptr;
    const bool want_hm = (keep_flags & IMPOP_KEEP_HAP_MAJOR) != 0;
    int rc = alloc_matrix(ctx, n_hap, n_site, want_hm, &m);
    if (rc) return rc;
    auto fail = [&](int code) {
        impop_matrix_free(ctx, m);
        return code;
    };
    const uint32_t wps = m->g.wps;
    std::vector<uint32_t> tables((size_t)(p->n_founder + 1) * wps, 0u);
    for (uint32_t h = 0; h < n_hap; ++h) {
        const uint32_t f = (uint32_t)(synth_hash(p->seed, 1000, h) % p->n_founder);
        tables[(size_t)f * wps + (h >> 5)] |= 1u << (h & 31);
        tables[(size_t)p->n_founder * wps + (h >> 5)] |= 1u << (h & 31);
    }
    void *d_tab = nullptr;
    rc = ctx_scratch(ctx, tables.size() * 4, &d_tab);
    if (rc) return fail(rc);
    hipError_t e = hipMemcpyAsync(d_tab, tables.data(), tables.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpyAsync(tables)", __FILE__, __LINE__));
    SynthDev sp;
    sp.seed = p->seed;
    sp.n_founder = p->n_founder;
    sp.thr_founder = (uint32_t)(p->p_founder * 4294967296.0);
    sp.thr_private = (uint32_t)(p->p_private_word * 4294967296.0);
    if (m->g.n_block) {
        const uint64_t grid = (m->g.n_block + 3) / 4;
        if (grid >= 0x7FFFFFFFull) {
            set_error("impop_matrix_synthetic: too many sites");
            return fail(IMPOP_E_INVALID);
        }
        hipLaunchKernelGGL(synth_sb_kernel, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, sp, (const uint32_t *)d_tab,
                           wps, m->g.G, m->g.r, m->g.n_block, n_site, site_begin, m->d_sb);
        if ((e = hipGetLastError()) != hipSuccess) return fail(hip_fail(e, "synth_sb_kernel", __FILE__, __LINE__));
        if (want_hm) {
            rc = launch_sb_to_hm(ctx, m->d_sb, m->g, 0, m->g.n_block, m->d_rb, 0, m->n_hap_pad, m->rb_nb, m->phi_row);
            if (rc) return fail(rc);
            rc = ensure_segmap(ctx, m);
            if (rc) return fail(rc);
        }
    }
    e = hipStreamSynchronize(ctx->stream);  // `tables` (pageable) must stay alive until the copy is done
    if (e != hipSuccess) return fail(hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
    *out = m;
    return IMPOP_OK;
}

IMPOP_API int impop_matrix_download(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                                    uint64_t *out, uint64_t row_stride_words) {
    REQUIRE(ctx && m, "impop_matrix_download: NULL argument");
    REQUIRE(site_begin <= site_end && site_end <= m->g.n_site, "impop_matrix_download: bad site range [%llu,%llu)",
            (unsigned long long)site_begin, (unsigned long long)site_end);
    const uint64_t W = site_end - site_begin;
    const uint64_t out_words = (W + 63) / 64;
    REQUIRE(row_stride_words >= out_words, "impop_matrix_download: row_stride_words too small");
    if (W == 0) return IMPOP_OK;
    REQUIRE(out, "impop_matrix_download: out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t b0 = site_begin / 64, b1 = (site_end + 63) / 64;
    const uint64_t nb = b1 - b0;
    const uint64_t stride = nb * 2;  // dwords per row in the temporary
    const uint32_t n_rows = m->g.n_hap;
    void *d_tmp = nullptr;
    int rc = ctx_scratch(ctx, (size_t)n_rows * stride * 4, &d_tmp);
    if (rc) return rc;
    rc = launch_sb_to_hm(ctx, m->d_sb, m->g, b0, b1, (uint32_t *)d_tmp, stride, n_rows);
    if (rc) return rc;
    std::vector<uint64_t> tmp((size_t)n_rows * nb);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d_tmp, tmp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint32_t sh = (uint32_t)(site_begin & 63);
    for (uint32_t i = 0; i < n_rows; ++i) {
        const uint64_t *src = tmp.data() + (size_t)i * nb;
        uint64_t *dst = out + (size_t)i * row_stride_words;
        for (uint64_t w = 0; w < out_words; ++w) {
            uint64_t v = src[w] >> sh;
            if (sh && w + 1 < nb) v |= src[w + 1] << (64 - sh);
            dst[w] = v;
        }
        if (W & 63) dst[out_words - 1] &= (~0ull) >> (64 - (W & 63));
    }
    return IMPOP_OK;
}

// ---- compaction to the variable sites ------------------------------------------------------------
// Monomorphic sites (c_s = 0 or c_s = n over ALL haplotypes) add 0 to every sum_s c(n - c) of every
// subset and are never segregating, so the scan statistics of a window depend on its variable sites
// and its LENGTH only.  A compacted matrix keeps those sites (with their original positions) and is
// scanned with windows in the original coordinates: same records, ~W/S times fewer bytes.
namespace impop {

// grid-stride over blocks: a workgroup per 4 blocks cost more in launches than in bytes (37 ms for 14.6 GB)
__global__ __launch_bounds__(256) void variable_mask_kernel(const uint32_t *__restrict__ sb, uint32_t wps, uint32_t G,
                                                            uint32_t r, uint64_t n_block, uint64_t n_site, uint32_t n_hap,
                                                            uint64_t *__restrict__ mask, uint32_t *__restrict__ cnt,
                                                            uint64_t *__restrict__ ones /* nullable: sites with c = n */) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    for (uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_block; b += stride) {
        const uint32_t *site = sb + b * 64ull * wps;
        uint32_t c = 0;
        for (uint32_t g = 0; g + 1 < G; ++g) {  // whole 16-byte granules: one coalesced 1 KiB wave load each
            const uint4 v = *reinterpret_cast<const uint4 *>(site + (uint64_t)g * 256 + lane * 4);
            c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
        for (uint32_t j = 0; j < r; ++j) c += __popc(site[(uint64_t)(G - 1) * 256 + lane * r + j]);
        const bool var = (b * 64 + lane < n_site) && c > 0 && c < n_hap;
        const uint64_t m = __ballot(var);
        const uint64_t o = ones ? __ballot((b * 64 + lane < n_site) && c == n_hap) : 0ull;
        if (lane == 0) {
            mask[b] = m; cnt[b] = (uint32_t)__popcll(m);
            if (ones) ones[b] = o;
        }
    }
}

constexpr uint32_t SCAN_CHUNK = 1024;  // blocks per workgroup in the two-level exclusive scan
__global__ __launch_bounds__(256) void chunk_sum_kernel(const uint32_t *__restrict__ cnt, uint64_t n, uint64_t *__restrict__ chunk_sum) {
    __shared__ uint64_t sh[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t t = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_CHUNK; i += 256)
        if (base + i < n) t += cnt[base + i];
    t = wave_sum_u64(t);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void chunk_scan_kernel(uint64_t *chunk_sum, uint64_t n_chunks, uint64_t *total) {  // one thread: n_chunks is small
    if (threadIdx.x || blockIdx.x) return;
    uint64_t run = 0;
    for (uint64_t i = 0; i < n_chunks; ++i) {
        const uint64_t v = chunk_sum[i];
        chunk_sum[i] = run;
        run += v;
    }
    *total = run;
}
__global__ __launch_bounds__(64) void block_base_kernel(const uint32_t *__restrict__ cnt, uint64_t n, const uint64_t *__restrict__ chunk_off,
                                                        uint64_t *__restrict__ base) {
    // one wave per chunk: 16 rounds of a 64-wide exclusive scan
    const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t run = chunk_off[blockIdx.x];
    for (uint32_t rnd = 0; rnd < SCAN_CHUNK / 64; ++rnd) {
        const uint64_t i = b0 + rnd * 64 + threadIdx.x;
        const uint64_t v = i < n ? cnt[i] : 0;
        uint64_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t o = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += o;
        }
        if (i < n) base[i] = run + incl - v;
        run += __shfl(incl, 63, 64);
    }
}
__global__ __launch_bounds__(256) void gather_variable_kernel(const uint32_t *__restrict__ sb, uint32_t wps, uint32_t G, uint32_t r,
                                                              uint64_t n_block, const uint64_t *__restrict__ mask,
                                                              const uint64_t *__restrict__ base, uint32_t *__restrict__ out,
                                                              uint64_t *__restrict__ pos) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    for (uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_block; b += stride) {
        const uint64_t m = mask[b];
        if (!m) continue;  // wave-uniform: most blocks of a real matrix hold a few variable sites, many none
        if (!((m >> lane) & 1ull)) continue;
        const uint64_t dest = base[b] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
        for (uint32_t k = 0; k < wps; ++k)
            out[sb_index(wps, G, r, dest >> 6, (uint32_t)(dest & 63), k)] = sb[sb_index(wps, G, r, b, lane, k)];
        pos[dest] = b * 64 + lane;
    }
}

// Two levels: a binary search over 10^7 positions (87 MB) misses the cache on nearly every step — 8192 window edges cost
// 0.7-4 ms per call, as much as the Gram launch they precede; every 4096th position (a few thousand entries, cache
// resident) narrows the search to one 32 KB run first.
__global__ void map_edges_kernel(const uint64_t *__restrict__ pos, uint64_t n_pos, const impop_window *__restrict__ win, uint64_t n_win,
                                 impop_window *__restrict__ out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * n_win) return;
    const uint64_t key = (e & 1) ? win[e >> 1].site_end : win[e >> 1].site_begin;
    uint64_t lo = 0, len = n_pos;  // first kept site at or right of `key` (= kept sites left of it)
    while (len) {
        const uint64_t half = len >> 1;
        if (pos[lo + half] < key) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    if (e & 1) out[e >> 1].site_end = lo;
    else { out[e >> 1].site_begin = lo; out[e >> 1].seq_len = win[e >> 1].seq_len; }
}

int map_windows_device(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n, std::vector<impop_window> &mapped) {
    if (!m->compact || !m->d_pos || n < 256) {  // few windows: the host search is quicker than a launch and two copies
        map_windows(m, windows, n, mapped);
        return IMPOP_OK;
    }
    REQUIRE((2 * n + 255) / 256 < 0x7FFFFFFFull, "map_windows_device: too many windows");
    void *d = nullptr;
    const int rc = ctx_aux(ctx, 0, 2 * n * sizeof(impop_window), &d);  // slot 0 (h-fst partials) is idle before the first launch of a call
    if (rc) return rc;
    impop_window *d_in = reinterpret_cast<impop_window *>(d), *d_out = d_in + n;
    mapped.resize(n);
    HIP_TRY(hipMemcpyAsync(d_in, windows, n * sizeof(impop_window), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(map_edges_kernel, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, m->d_pos, (uint64_t)m->pos.size(),
                       d_in, n, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mapped.data(), d_out, n * sizeof(impop_window), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

uint64_t pos_lower_bound(const impop_matrix *m, uint64_t s) {
    const std::vector<uint64_t> &pos = m->pos, &co = m->pos_coarse;
    if (co.empty()) return (uint64_t)(std::lower_bound(pos.begin(), pos.end(), s) - pos.begin());
    // first coarse entry >= s: the answer lies in (its predecessor's index, its index]
    const size_t c = (size_t)(std::lower_bound(co.begin(), co.end(), s) - co.begin());
    const size_t lo = c == 0 ? 0 : ((c - 1) << POS_COARSE_SHIFT) + 1;
    const size_t hi = c < co.size() ? (c << POS_COARSE_SHIFT) : pos.size();
    return (uint64_t)(std::lower_bound(pos.begin() + lo, pos.begin() + hi, s) - pos.begin());
}

// The window edges of a batch, 32 searches in lock step: after the coarse level every search is a dozen DEPENDENT cache
// misses in its own 32 KB run (250 ns per edge one after the other: 2 ms per 4096 windows, as long as the Gram launch that
// follows); interleaved, the misses of different edges overlap, and each step's next probe is prefetched while the other
// edges take theirs.
void map_windows(const impop_matrix *m, const impop_window *windows, uint64_t n, std::vector<impop_window> &mapped) {
    mapped.assign(windows, windows + n);
    if (!m->compact) return;
    const uint64_t *pos = m->pos.data();
    const std::vector<uint64_t> &co = m->pos_coarse;
    constexpr int Q = 32;
    uint64_t key[Q];
    size_t base[Q], len[Q];
    for (uint64_t i0 = 0; i0 < 2 * n; i0 += Q) {
        const int cnt = (int)std::min<uint64_t>(Q, 2 * n - i0);
        for (int q = 0; q < cnt; ++q) {
            const uint64_t e = i0 + q;
            key[q] = (e & 1) ? windows[e >> 1].site_end : windows[e >> 1].site_begin;
            size_t lo = 0, hi = m->pos.size();
            if (!co.empty()) {  // first coarse entry >= key: the answer lies in (its predecessor's index, its index]
                const size_t c = (size_t)(std::lower_bound(co.begin(), co.end(), key[q]) - co.begin());
                lo = c == 0 ? 0 : ((c - 1) << POS_COARSE_SHIFT) + 1;
                hi = c < co.size() ? (c << POS_COARSE_SHIFT) : m->pos.size();
            }
            base[q] = lo;
            len[q] = hi - lo;
            if (len[q]) __builtin_prefetch(pos + lo + (len[q] >> 1));
        }
        for (bool more = true; more;) {
            more = false;
            for (int q = 0; q < cnt; ++q) {
                if (!len[q]) continue;
                const size_t half = len[q] >> 1;
                if (pos[base[q] + half] < key[q]) {
                    base[q] += half + 1;
                    len[q] -= half + 1;
                } else {
                    len[q] = half;
                }
                if (len[q]) {
                    __builtin_prefetch(pos + base[q] + (len[q] >> 1));
                    more = true;
                }
            }
        }
        for (int q = 0; q < cnt; ++q) {
            const uint64_t e = i0 + q;
            if (e & 1) mapped[e >> 1].site_end = base[q];
            else mapped[e >> 1].site_begin = base[q];
        }
    }
}

}  // namespace impop

IMPOP_API int impop_matrix_compact(impop_ctx *ctx, const impop_matrix *in, impop_matrix **out) {
    REQUIRE(ctx && in && out, "impop_matrix_compact: NULL argument");
    *out = nullptr;
    REQUIRE(!in->compact, "impop_matrix_compact: matrix is already compacted");
    HIP_TRY(hipSetDevice(ctx->device));
    const SbGeom &g = in->g;
    const uint64_t nb = g.n_block, n_chunks = (nb + SCAN_CHUNK - 1) / SCAN_CHUNK;
    REQUIRE((nb + 3) / 4 < 0x7FFFFFFFull, "impop_matrix_compact: matrix too long for one launch");
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_mask = 0, o_cnt = o_mask + up(nb * 8), o_base = o_cnt + up(nb * 4), o_chunk = o_base + up(nb * 8),
                 o_total = o_chunk + up(n_chunks * 8);
    void *d = nullptr;
    int rc = ctx_scratch(ctx, o_total + 256, &d);
    if (rc) return rc;
    uint64_t *d_mask = (uint64_t *)((char *)d + o_mask), *d_base = (uint64_t *)((char *)d + o_base),
             *d_chunk = (uint64_t *)((char *)d + o_chunk), *d_total = (uint64_t *)((char *)d + o_total);
    uint32_t *d_cnt = (uint32_t *)((char *)d + o_cnt);
    uint64_t n_kept = 0;
    // a source that kept its hap-major copy hands the all-pairs path on: the compacted matrix gets its own RB32
    // operand and the bitmap of the dropped all-ones sites (their count — for a weighted source the sum of their
    // weights, from host prefix sums built below — comes back as a per-window constant)
    const bool want_pairs = in->d_rb != nullptr;
    uint64_t *d_ones = nullptr;
    if (want_pairs && nb) HIP_TRY(hipMalloc((void **)&d_ones, nb * 8 + 256));
    if (nb) {
        const uint32_t wide_grid = (uint32_t)std::min<uint64_t>((nb + 3) / 4, 32ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256));
        hipLaunchKernelGGL(variable_mask_kernel, dim3(wide_grid), dim3(256), 0, ctx->stream, in->d_sb, g.wps, g.G, g.r, nb, g.n_site,
                           g.n_hap, d_mask, d_cnt, d_ones);
        hipLaunchKernelGGL(chunk_sum_kernel, dim3((uint32_t)n_chunks), dim3(256), 0, ctx->stream, d_cnt, nb, d_chunk);
        hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(64), 0, ctx->stream, d_chunk, n_chunks, d_total);
        hipLaunchKernelGGL(block_base_kernel, dim3((uint32_t)n_chunks), dim3(64), 0, ctx->stream, d_cnt, nb, d_chunk, d_base);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&n_kept, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    impop_matrix *m = nullptr;
    rc = alloc_matrix(ctx, g.n_hap, n_kept, want_pairs, &m);
    if (rc) {
        if (d_ones) hipFree(d_ones);
        return rc;
    }
    m->d_onesmap = reinterpret_cast<uint32_t *>(d_ones);  // uint64 per 64-site block == two dwords of 32 sites
    auto fail = [&](int code) {
        impop_matrix_free(ctx, m);
        return code;
    };
    m->compact = true;
    m->n_site_orig = g.n_site;
    m->pos.resize(n_kept);
    if (n_kept) {
        uint64_t *d_pos = nullptr;
        hipError_t e = hipMalloc((void **)&d_pos, n_kept * 8);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc(positions)", __FILE__, __LINE__));
        auto fail2 = [&](hipError_t err, const char *what) {
            hipFree(d_pos);
            return fail(hip_fail(err, what, __FILE__, __LINE__));
        };
        if ((e = hipMemsetAsync(m->d_sb, 0, m->sb_bytes, ctx->stream)) != hipSuccess) return fail2(e, "hipMemsetAsync");
        const uint32_t wide_grid = (uint32_t)std::min<uint64_t>((nb + 3) / 4, 32ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256));
        hipLaunchKernelGGL(gather_variable_kernel, dim3(wide_grid), dim3(256), 0, ctx->stream, in->d_sb, g.wps, g.G, g.r, nb, d_mask,
                           d_base, m->d_sb, d_pos);
        if ((e = hipGetLastError()) != hipSuccess) return fail2(e, "gather_variable_kernel");
        if ((e = hipMemcpyAsync(m->pos.data(), d_pos, n_kept * 8, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
            return fail2(e, "hipMemcpyAsync(positions)");
        if (want_pairs) {
            const int rc2 = launch_sb_to_hm(ctx, m->d_sb, m->g, 0, m->g.n_block, m->d_rb, 0, m->n_hap_pad, m->rb_nb, m->phi_row);
            if (rc2) { hipFree(d_pos); return fail(rc2); }
        }
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail2(e, "hipStreamSynchronize");
        m->d_pos = d_pos;  // kept: window edges are mapped on the device (map_windows_device)
    }
    m->pos_coarse.clear();
    for (size_t k = 0; k < m->pos.size(); k += (size_t)1 << POS_COARSE_SHIFT) m->pos_coarse.push_back(m->pos[k]);
    if (!in->wt_prefix.empty()) {
        // weighted input: the kept columns keep their weights; a window's W is still the sum over ALL its original
        // columns (monomorphic ones included), so the prefix sums of the original weights travel along
        m->wt_prefix = in->wt_prefix;
        if (n_kept) {
            std::vector<uint32_t> kept(n_kept);
            for (uint64_t k = 0; k < n_kept; ++k) kept[k] = (uint32_t)(in->wt_prefix[m->pos[k] + 1] - in->wt_prefix[m->pos[k]]);
            hipError_t e;
            if ((e = hipMalloc((void **)&m->d_wt, n_kept * 4ull)) != hipSuccess) return fail(hip_fail(e, "hipMalloc(weights)", __FILE__, __LINE__));
            if ((e = hipMemcpy(m->d_wt, kept.data(), n_kept * 4ull, hipMemcpyHostToDevice)) != hipSuccess)
                return fail(hip_fail(e, "hipMemcpy(weights)", __FILE__, __LINE__));
            if (want_pairs) {
                m->kept_wt_prefix.resize(n_kept + 1);
                m->kept_wt_prefix[0] = 0;
                for (uint64_t k = 0; k < n_kept; ++k) m->kept_wt_prefix[k + 1] = m->kept_wt_prefix[k] + kept[k];
            }
        }
        if (want_pairs && m->kept_wt_prefix.empty()) m->kept_wt_prefix.assign(1, 0);  // nothing kept
        if (want_pairs) {  // weights of the dropped all-ones sites, as prefix sums over the original coordinates
            std::vector<uint64_t> ones(nb);
            hipError_t e;
            if (nb && (e = hipMemcpy(ones.data(), d_ones, nb * 8, hipMemcpyDeviceToHost)) != hipSuccess)
                return fail(hip_fail(e, "hipMemcpy(all-ones bitmap)", __FILE__, __LINE__));
            m->ones_wt_prefix.resize(g.n_site + 1);
            uint64_t acc = 0;
            for (uint64_t i = 0; i < g.n_site; ++i) {
                m->ones_wt_prefix[i] = acc;
                if ((ones[i >> 6] >> (i & 63)) & 1ull) acc += in->wt_prefix[i + 1] - in->wt_prefix[i];
            }
            m->ones_wt_prefix[g.n_site] = acc;
        }
    }
    *out = m;
    return IMPOP_OK;
}

namespace impop {
void matrix_drop_derived(const impop_matrix *m) {
    if (m->d_wplanes) hipFree(m->d_wplanes);
    if (m->d_rb_masked) hipFree(m->d_rb_masked);
    if (m->d_segmap) hipFree(m->d_segmap);
    m->d_wplanes = nullptr; m->d_rb_masked = nullptr; m->d_segmap = nullptr;
    m->wplane_bits = 0; m->wplane_stride = 0;
}
}  // namespace impop

IMPOP_API int impop_matrix_set_site_weights(impop_ctx *ctx, impop_matrix *m, const uint32_t *weights_host) {
    REQUIRE(ctx && m, "impop_matrix_set_site_weights: NULL argument");
    NOT_COMPACT(m, "impop_matrix_set_site_weights");
    REQUIRE(m->users == 0, "impop_matrix_set_site_weights: %d scan plan(s) were built with the current weights; destroy them first",
            m->users);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // launches in flight may still read the old weights
    if (m->d_wt) {
        HIP_TRY(hipFree(m->d_wt));
        m->d_wt = nullptr;
    }
    matrix_drop_derived(m);  // weight planes / masked operand belong to the old weights
    m->wt_prefix.clear();
    if (!weights_host) return IMPOP_OK;
    // host prefix sums: a window's W = sum of its columns' weights, looked up when a plan is built
    m->wt_prefix.resize(m->g.n_site + 1);
    m->wt_prefix[0] = 0;
    for (uint64_t i = 0; i < m->g.n_site; ++i) m->wt_prefix[i + 1] = m->wt_prefix[i] + weights_host[i];
    if (m->g.n_site == 0) return IMPOP_OK;
    HIP_TRY(hipMalloc((void **)&m->d_wt, m->g.n_site * 4ull));
    HIP_TRY(hipMemcpy(m->d_wt, weights_host, m->g.n_site * 4ull, hipMemcpyHostToDevice));
    return IMPOP_OK;
}

IMPOP_API int impop_matrix_positions(const impop_matrix *m, uint64_t first, uint64_t count, uint64_t *out, uint64_t *n_site_orig) {
    REQUIRE(m, "impop_matrix_positions: matrix is NULL");
    REQUIRE(m->compact, "impop_matrix_positions: not a compacted matrix");
    REQUIRE(first <= m->pos.size() && count <= m->pos.size() - first, "impop_matrix_positions: range out of bounds");
    REQUIRE(count == 0 || out, "impop_matrix_positions: out is NULL");
    for (uint64_t i = 0; i < count; ++i) out[i] = m->pos[first + i];
    if (n_site_orig) *n_site_orig = m->n_site_orig;
    return IMPOP_OK;
}

IMPOP_API int impop_matrix_info(const impop_matrix *m, uint32_t *n_hap, uint64_t *n_site, uint64_t *device_bytes,
                                uint32_t *bytes_per_site) {
    REQUIRE(m, "impop_matrix_info: matrix is NULL");
    if (n_hap) *n_hap = m->g.n_hap;
    if (n_site) *n_site = m->g.n_site;
    if (device_bytes) *device_bytes = m->sb_bytes + m->rb_bytes;
    if (bytes_per_site) *bytes_per_site = m->g.wps * 4;
    return IMPOP_OK;
}

IMPOP_API int impop_matrix_free(impop_ctx *ctx, impop_matrix *m) {
    if (!m) return IMPOP_OK;
    REQUIRE(m->users == 0, "impop_matrix_free: %d scan plan(s) still reference this matrix; destroy them first", m->users);
    if (ctx) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
    }
    if (m->d_sb) hipFree(m->d_sb);
    if (m->d_rb) hipFree(m->d_rb);
    if (m->d_wt) hipFree(m->d_wt);
    if (m->d_onesmap) hipFree(m->d_onesmap);
    if (m->d_pos) hipFree(m->d_pos);
    matrix_drop_derived(m);
    delete m;
    return IMPOP_OK;
}
