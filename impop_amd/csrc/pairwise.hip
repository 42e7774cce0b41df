// pairwise.hip — the all-pairs path: I_ij = sum_s b_is b_js for every haplotype pair of a
// window, then the full pica2 / h-fst semantics (thresholds, rounding,
// greedy grouping) on identities formed on the fly from the integer Gram matrix.
//
// This path is VALU-bound, not HBM-bound: its roof is the popcount issue rate
// (2 lane-ops per haplotype pair per 32 sites), reported separately from the scan.
#include <vector>

#include "stats_kernels.h"

namespace impop {

// ---- Gram kernel v2 ---------------------------------------------------------------------
// 128 x 128 haplotype tile per 256-thread workgroup (upper-triangular tile grid x windows);
// thread (ty, tx) owns the 8 x 8 outputs rows {ty+16r} x cols {tx+16c}.  The site axis is
// consumed in 16-dword (512-site) chunks staged through LDS with row pitch 20 dwords: the
// 16 distinct rows a wave touches per ds_read_b128 then start on distinct 4-bank groups
// (tx*20 mod 64 = 0,20,40,60,16,...: all different multiples of 4) => conflict-free.  Per
// chunk a thread issues 4x16 ds_read_b128 for 8*8*16 AND+BCNT pairs (1:32 LDS:VALU), so the
// kernel is bound by the VALU popcount rate.  Global->LDS staging is double-buffered through
// registers (loads for chunk t+1 are issued before the math of chunk t).  Diagonal tiles skip
// the r > c sub-blocks (strictly below the diagonal for every thread), 28 of 64.
constexpr int GT = 128;      // tile edge (haplotypes)
constexpr int KC = 16;       // dwords per K-chunk
constexpr int PITCH = 20;    // LDS row pitch (dwords)

struct GramWindow {
    uint64_t site_begin, site_end;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool DIAG>
__device__ __forceinline__ void gram_chunk_math(const uint32_t *__restrict__ As, const uint32_t *__restrict__ Bs,
                                                uint32_t ty, uint32_t tx, int32_t (&acc)[8][8]) {
#pragma unroll 1
    for (int k4 = 0; k4 < KC / 4; ++k4) {
        u32x4 a[8], b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = *reinterpret_cast<const u32x4 *>(As + (ty + 16 * r) * PITCH + 4 * k4);
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = *reinterpret_cast<const u32x4 *>(Bs + (tx + 16 * c) * PITCH + 4 * k4);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (DIAG && r > c) continue;
                acc[r][c] += __popc(a[r].x & b[c].x) + __popc(a[r].y & b[c].y) + __popc(a[r].z & b[c].z) +
                             __popc(a[r].w & b[c].w);
            }
    }
}

__global__ __launch_bounds__(256, 2) void gram_kernel(const uint32_t *__restrict__ hm, uint64_t hm_stride, uint32_t n_tiles,
                                                   const GramWindow *__restrict__ wins, int32_t *__restrict__ out,
                                                   uint32_t ld, uint64_t out_stride) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][2][GT * PITCH];  // [buffer][A|B]
    // decode upper-triangular tile pair (ti <= tj) from blockIdx.x
    uint32_t rem = blockIdx.x, ti = 0;
    while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ++ti; }
    const uint32_t tj = ti + rem;
    const bool diag = ti == tj;
    const GramWindow w = wins[blockIdx.y];
    const uint32_t tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    int32_t acc[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    if (w.site_end > w.site_begin) {
        const uint64_t d0 = w.site_begin >> 5, d1 = (w.site_end + 31) >> 5;  // dword range of the window
        const uint64_t c0 = d0 & ~3ull;                                        // 16-byte aligned start
        const uint32_t first_mask = 0xFFFFFFFFu << (w.site_begin & 31);
        const uint32_t last_mask = (w.site_end & 31) ? (0xFFFFFFFFu >> (32 - (w.site_end & 31))) : 0xFFFFFFFFu;
        // staging map: 128 rows x 4 uint4 per tile = 512 uint4; thread t moves uint4 #t and #t+256
        const uint32_t srow = tid >> 2, scol = (tid & 3) * 4;  // rows srow and srow+64
        const uint32_t *gA = hm + (uint64_t)(ti * GT + srow) * hm_stride + scol;
        const uint32_t *gB = hm + (uint64_t)(tj * GT + srow) * hm_stride + scol;
        const uint64_t row64 = 64 * hm_stride;
        auto mask_of = [&](uint64_t d) -> uint32_t {
            uint32_t m = (d >= d0 && d < d1) ? 0xFFFFFFFFu : 0u;
            if (d == d0) m &= first_mask;
            if (d == d1 - 1) m &= last_mask;
            return m;
        };
        u32x4 ra0, ra1, rb0, rb1;
        auto fetch = [&](uint64_t dk) {
            // rows are padded to a multiple of 4 dwords and cover every block of the matrix, so a
            // 16-byte load at an aligned dword < hm_stride is always in bounds
            const bool in = dk + scol < hm_stride;
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra0 = in ? *reinterpret_cast<const u32x4 *>(gA + dk) : z;
            ra1 = in ? *reinterpret_cast<const u32x4 *>(gA + row64 + dk) : z;
            rb0 = in ? *reinterpret_cast<const u32x4 *>(gB + dk) : z;
            rb1 = in ? *reinterpret_cast<const u32x4 *>(gB + row64 + dk) : z;
            const uint64_t d = dk + scol;
            const u32x4 m = {mask_of(d), mask_of(d + 1), mask_of(d + 2), mask_of(d + 3)};
            ra0 &= m; ra1 &= m;  // masking one operand suffices for AND
        };
        auto stash = [&](int buf) {
            *reinterpret_cast<u32x4 *>(&lds[buf][0][srow * PITCH + scol]) = ra0;
            *reinterpret_cast<u32x4 *>(&lds[buf][0][(srow + 64) * PITCH + scol]) = ra1;
            *reinterpret_cast<u32x4 *>(&lds[buf][1][srow * PITCH + scol]) = rb0;
            *reinterpret_cast<u32x4 *>(&lds[buf][1][(srow + 64) * PITCH + scol]) = rb1;
        };
        fetch(c0);
        stash(0);
        __syncthreads();
        int buf = 0;
        for (uint64_t dk = c0; dk < d1; dk += KC) {
            const bool more = dk + KC < d1;
            if (more) fetch(dk + KC);
            if (diag) gram_chunk_math<true>(lds[buf][0], lds[buf][1], ty, tx, acc);
            else gram_chunk_math<false>(lds[buf][0], lds[buf][1], ty, tx, acc);
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    int32_t *o = out + (uint64_t)blockIdx.y * out_stride;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (diag && r > c) continue;  // lower half of a diagonal tile: filled by symmetry on export
            o[(uint64_t)(ti * GT + ty + 16 * r) * ld + (tj * GT + tx + 16 * c)] = acc[r][c];
        }
}

// ---- Gram kernel v3: int8 MFMA ----------------------------------------------------------------
// G[i][j] = sum_s x_i[s] x_j[s] as a dense contraction on the matrix cores: bits are expanded to
// 0/1 bytes in LDS and multiplied with v_mfma_i32_32x32x32_i8 (exact: int32 accumulation).
// Measured motivation (tools/micro/valu_rate.hip): v_bcnt_u32_b32 is a half-rate instruction on
// gfx950 (4.8 vs 2.55 cycles per wave-instruction for v_and), so the VALU popcount form tops out
// near 1e5 windows/s at n = 465, W = 50 000, while int8 MFMA has ~10x the raw MAC rate.
//
// Workgroup = 128 x 128 haplotype tile, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles of 32 x 32.
// The site axis is consumed in 512-site super-chunks (16 dwords per row, prefetched one ahead in
// registers); thread (r = tid/2, h = tid%2) owns dwords 8h..8h+7 of row r of BOTH operand tiles.
// A sub-chunk (K = 64) takes dword j of h = 0 and dword j of h = 1: the sites inside a sub-chunk
// are not contiguous, which is irrelevant for a sum over sites, and it keeps every lane busy in
// the expansion.  Byte rows in LDS have pitch 80 B so that ds_read_b128 fragment reads of 16
// consecutive rows start in distinct 16-byte bank groups.  Both MFMA operands use the same
// (lane>>5, byte) -> site mapping, so the result does not depend on the instruction's internal
// k order; the C/D map is the documented col = lane&31, row = (reg&3) + 8(reg>>2) + 4(lane>>5).
constexpr int BP = 80;  // LDS byte-row pitch

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// 4 bits -> 4 bytes of 0/1: (nibble * 0x204081) & 0x01010101 (bit i lands on bit 8i)
__device__ __forceinline__ uint32_t expand_nibble(uint32_t x, int i) {
    return __umul24((x >> (4 * i)) & 0xFFu, 0x204081u) & 0x01010101u;  // bits 4..7 of the byte land off the kept positions
}
__device__ __forceinline__ void expand_dword_to_lds(uint32_t x, unsigned char *dst /*32 bytes, 16-B aligned*/) {
    u32x4 lo = {expand_nibble(x, 0), expand_nibble(x, 1), expand_nibble(x, 2), expand_nibble(x, 3)};
    u32x4 hi = {expand_nibble(x, 4), expand_nibble(x, 5), expand_nibble(x, 6), expand_nibble(x, 7)};
    *reinterpret_cast<u32x4 *>(dst) = lo;
    *reinterpret_cast<u32x4 *>(dst + 16) = hi;
}

__global__ __launch_bounds__(256, 2) void gram_mfma_kernel(const uint32_t *__restrict__ hm, uint64_t hm_stride,
                                                           uint32_t n_tiles, const GramWindow *__restrict__ wins,
                                                           int32_t *__restrict__ out, uint32_t ld, uint64_t out_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][GT * BP];  // [buffer][A|B] byte tiles, 40 KB
    uint32_t rem = blockIdx.x, ti = 0;
    while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ++ti; }
    const uint32_t tj = ti + rem;
    const bool diag = ti == tj;
    const GramWindow w = wins[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const bool wave_active = !(diag && wr == 1 && wc == 0);  // strictly below the diagonal: nothing to compute
    i32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    if (w.site_end > w.site_begin) {
        const uint64_t d0 = w.site_begin >> 5, d1 = (w.site_end + 31) >> 5;
        const uint64_t c0 = d0 & ~3ull;
        const uint32_t first_mask = 0xFFFFFFFFu << (w.site_begin & 31);
        const uint32_t last_mask = (w.site_end & 31) ? (0xFFFFFFFFu >> (32 - (w.site_end & 31))) : 0xFFFFFFFFu;
        const uint32_t r = tid >> 1, h = tid & 1;
        const uint32_t *gA = hm + (uint64_t)(ti * GT + r) * hm_stride + 8 * h;
        const uint32_t *gB = hm + (uint64_t)(tj * GT + r) * hm_stride + 8 * h;
        auto mask_of = [&](uint64_t d) -> uint32_t {
            uint32_t m = (d >= d0 && d < d1) ? 0xFFFFFFFFu : 0u;
            if (d == d0) m &= first_mask;
            if (d == d1 - 1) m &= last_mask;
            return m;
        };
        u32x4 ca0, ca1, cb0, cb1, na0, na1, nb0, nb1;  // current / next super-chunk: 8 dwords of A and of B
        auto fetch = [&](uint64_t dk) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            const uint64_t d = dk + 8 * h;
            const bool in0 = d < hm_stride, in1 = d + 4 < hm_stride;
            na0 = in0 ? *reinterpret_cast<const u32x4 *>(gA + dk) : z;
            na1 = in1 ? *reinterpret_cast<const u32x4 *>(gA + dk + 4) : z;
            nb0 = in0 ? *reinterpret_cast<const u32x4 *>(gB + dk) : z;
            nb1 = in1 ? *reinterpret_cast<const u32x4 *>(gB + dk + 4) : z;
            const u32x4 m0 = {mask_of(d), mask_of(d + 1), mask_of(d + 2), mask_of(d + 3)};
            const u32x4 m1 = {mask_of(d + 4), mask_of(d + 5), mask_of(d + 6), mask_of(d + 7)};
            na0 &= m0; na1 &= m1;  // a masked A byte is 0 => the product is 0
        };
        unsigned char *const wrA = &lds[0][0][r * BP + 32 * h], *const wrB = &lds[0][1][r * BP + 32 * h];
        constexpr int BUF = 2 * GT * BP;  // bytes between the two buffers
        const uint32_t frag_off = (lane & 31) * BP + 16 * (lane >> 5);
        fetch(c0);
        for (uint64_t dk = c0; dk < d1; dk += 16) {
            ca0 = na0; ca1 = na1; cb0 = nb0; cb1 = nb1;
            if (dk + 16 < d1) fetch(dk + 16);
            // sub-chunk 0 of this super-chunk
            expand_dword_to_lds(ca0.x, wrA);
            expand_dword_to_lds(cb0.x, wrB);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int buf = j & 1;
                if (j < 7) {  // expand sub-chunk j+1 into the other buffer while this one feeds the MFMAs
                    const uint32_t xa = (j + 1) < 4 ? ca0[(j + 1) & 3] : ca1[(j + 1) & 3];
                    const uint32_t xb = (j + 1) < 4 ? cb0[(j + 1) & 3] : cb1[(j + 1) & 3];
                    expand_dword_to_lds(xa, wrA + (buf ^ 1) * BUF);
                    expand_dword_to_lds(xb, wrB + (buf ^ 1) * BUF);
                }
                if (wave_active) {
                    const unsigned char *pa = &lds[buf][0][(64 * wr) * BP + frag_off];
                    const unsigned char *pb = &lds[buf][1][(64 * wc) * BP + frag_off];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const i32x4 a0 = *reinterpret_cast<const i32x4 *>(pa + 32 * ks);
                        const i32x4 a1 = *reinterpret_cast<const i32x4 *>(pa + 32 * BP + 32 * ks);
                        const i32x4 b0 = *reinterpret_cast<const i32x4 *>(pb + 32 * ks);
                        const i32x4 b1 = *reinterpret_cast<const i32x4 *>(pb + 32 * BP + 32 * ks);
                        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    }
    if (wave_active) {
        int32_t *o = out + (uint64_t)blockIdx.y * out_stride;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t row = ti * GT + 64 * wr + 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const uint32_t col = tj * GT + 64 * wc + 32 * nt + (lane & 31);
                    o[(uint64_t)row * ld + col] = acc[mt][nt][e];
                }
    }
}

#ifndef IMPOP_GRAM_MFMA
#define IMPOP_GRAM_MFMA 1  // 0: VALU AND+BCNT kernel (kept for A/B measurements)
#endif

// mirror the upper tiles into the lower triangle (only for host export)
__global__ void gram_symmetrize_kernel(int32_t *g, uint32_t ld) {
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ld && j < ld && i > j) g[(uint64_t)i * ld + j] = g[(uint64_t)j * ld + i];
}

__global__ void identity_dense_kernel(SimBatch b, uint32_t n, double *__restrict__ out) {
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || j >= n) return;
    const SimView S = sim_view(b, 0);
    out[(uint64_t)i * n + j] = sim_get(S, i, j);
}

struct PairFinalIn {
    const Pica2Out *pica;
    const HfstOut *hfst;
    const impop_window_stats *scan;  // integer S / W from the site scan of the same windows
};
__global__ void pairwise_finalize_kernel(PairFinalIn in, uint64_t n_windows, uint32_t nP, int d_pi_mode, int s_scope,
                                         impop_pairwise_stats *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_windows) return;
    const Pica2Out p = in.pica[i];
    const HfstOut h = in.hfst[i];
    const impop_window_stats s = in.scan[i];
    impop_pairwise_stats r;
    r.pi = p.pi; r.pi_site = p.pi_site;
    r.fst = h.v[0]; r.pi_a = h.v[1]; r.pi_b = h.v[2]; r.pi_xy = h.v[3]; r.dxy = h.v[4]; r.da = h.v[5];
    r.n_groups = p.n_groups; r.s_all = s.s_all; r.s_p = s.s_p; r.n_sites = s.n_sites; r.reserved = 0;
    const double S = (double)(s_scope == 0 ? s.s_all : s.s_p);
    const double pin = d_pi_mode == 0 ? py_round(p.pi_site, 8) : d_pi_mode == 1 ? p.pi_site : p.pi * (double)s.n_sites;
    double D = __builtin_nan("");
    if (nP >= 2 && pin == pin && pin >= 0) {
        const TajConsts c = tajima_consts((int64_t)nP);
        D = tajima_d_from(c, S, pin, nullptr, nullptr);
    }
    r.tajima_d = D;
    out[i] = r;
}

static int launch_gram(impop_ctx *ctx, const impop_matrix *m, const GramWindow *d_wins, uint32_t n_win, int32_t *d_out) {
    const uint32_t T = m->n_hap_pad / GT;
    const uint32_t pairs = T * (T + 1) / 2;
    REQUIRE(n_win <= 65535, "gram: at most 65535 windows per launch");
#if IMPOP_GRAM_MFMA
    hipLaunchKernelGGL(gram_mfma_kernel, dim3(pairs, n_win), dim3(256), 0, ctx->stream, m->d_hm, m->hm_stride, T, d_wins,
                       d_out, m->n_hap_pad, (uint64_t)m->n_hap_pad * m->n_hap_pad);
#else
    hipLaunchKernelGGL(gram_kernel, dim3(pairs, n_win), dim3(256), 0, ctx->stream, m->d_hm, m->hm_stride, T, d_wins, d_out,
                       m->n_hap_pad, (uint64_t)m->n_hap_pad * m->n_hap_pad);
#endif
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

struct Carve2 {
    char *base;
    size_t off = 0;
    explicit Carve2(void *p) : base((char *)p) {}
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) / 256 * 256;
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

}  // namespace impop

using namespace impop;

static int check_pairwise_args(impop_ctx *ctx, const impop_matrix *m, uint64_t s0, uint64_t s1, const char *fn) {
    REQUIRE(ctx && m, "%s: NULL argument", fn);
    REQUIRE(m->d_hm, "%s: matrix was created without IMPOP_KEEP_HAP_MAJOR", fn);
    REQUIRE(s0 <= s1 && s1 <= m->g.n_site, "%s: bad site range [%llu,%llu)", fn, (unsigned long long)s0,
            (unsigned long long)s1);
    REQUIRE(s1 - s0 < (1ull << 31), "%s: window longer than 2^31 sites overflows int32 counts", fn);
    return IMPOP_OK;
}

IMPOP_API int impop_pairwise_counts(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                                    int32_t *out_host) {
    int rc = check_pairwise_args(ctx, m, site_begin, site_end, "impop_pairwise_counts");
    if (rc) return rc;
    REQUIRE(out_host, "impop_pairwise_counts: out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = m->g.n_hap, ld = m->n_hap_pad;
    void *d = nullptr;
    rc = ctx_scratch(ctx, 512 + (size_t)ld * ld * 4, &d);
    if (rc) return rc;
    Carve2 cv(d);
    GramWindow *d_w = cv.take<GramWindow>(1);
    int32_t *d_g = cv.take<int32_t>((size_t)ld * ld);
    GramWindow w{site_begin, site_end};
    HIP_TRY(hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gram(ctx, m, d_w, 1, d_g);
    if (rc) return rc;
    hipLaunchKernelGGL(gram_symmetrize_kernel, dim3((ld + 15) / 16, (ld + 15) / 16), dim3(16, 16), 0, ctx->stream, d_g, ld);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(out_host, (size_t)n * 4, d_g, (size_t)ld * 4, (size_t)n * 4, n, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_pairwise_identity(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                                      int identity_kind, double *out_host) {
    int rc = check_pairwise_args(ctx, m, site_begin, site_end, "impop_pairwise_identity");
    if (rc) return rc;
    REQUIRE(out_host, "impop_pairwise_identity: out is NULL");
    REQUIRE(identity_kind == IMPOP_IDENTITY_MATCH || identity_kind == IMPOP_IDENTITY_DICE,
            "impop_pairwise_identity: unknown identity kind %d", identity_kind);
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = m->g.n_hap, ld = m->n_hap_pad;
    void *d = nullptr;
    rc = ctx_scratch(ctx, 1024 + (size_t)ld * ld * 4 + (size_t)n * n * 8, &d);
    if (rc) return rc;
    Carve2 cv(d);
    GramWindow *d_w = cv.take<GramWindow>(1);
    uint64_t *d_W = cv.take<uint64_t>(1);
    int32_t *d_g = cv.take<int32_t>((size_t)ld * ld);
    double *d_id = cv.take<double>((size_t)n * n);
    GramWindow w{site_begin, site_end};
    const uint64_t W = site_end - site_begin;
    HIP_TRY(hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_W, &W, 8, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gram(ctx, m, d_w, 1, d_g);
    if (rc) return rc;
    SimBatch b{};
    b.gram = d_g; b.stride = (uint64_t)ld * ld; b.ld = ld; b.n = n; b.W = d_W; b.kind = identity_kind; b.round_digits = -1;
    hipLaunchKernelGGL(identity_dense_kernel, dim3((n + 15) / 16, (n + 15) / 16), dim3(16, 16), 0, ctx->stream, b, n, d_id);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, d_id, (size_t)n * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_pairwise_scan(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                                  const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b,
                                  const impop_pairwise_params *params, impop_pairwise_stats *out_host) {
    REQUIRE(ctx && m && params, "impop_pairwise_scan: NULL argument");
    REQUIRE(params->struct_size == sizeof(impop_pairwise_params), "impop_pairwise_params.struct_size mismatch");
    REQUIRE(m->d_hm, "impop_pairwise_scan: matrix was created without IMPOP_KEEP_HAP_MAJOR");
    REQUIRE(params->identity_kind == IMPOP_IDENTITY_MATCH || params->identity_kind == IMPOP_IDENTITY_DICE,
            "impop_pairwise_scan: unknown identity kind");
    REQUIRE(params->round_digits <= 19, "impop_pairwise_scan: round_digits > 19 unsupported");
    REQUIRE(params->d_pi_mode >= 0 && params->d_pi_mode <= 2 && (params->s_scope == 0 || params->s_scope == 1),
            "impop_pairwise_scan: bad d_pi_mode / s_scope");
    if (!n_windows) return IMPOP_OK;
    REQUIRE(windows && out_host, "impop_pairwise_scan: NULL windows/out");
    for (uint64_t i = 0; i < n_windows; ++i) {
        int rc = check_pairwise_args(ctx, m, windows[i].site_begin, windows[i].site_end, "impop_pairwise_scan");
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t n = m->g.n_hap, ld = m->n_hap_pad;
    // integer S / W of the same windows from the streaming scan
    impop_scan_params sp;
    sp.struct_size = sizeof sp; sp.d_pi_mode = params->d_pi_mode; sp.s_scope = params->s_scope; sp.tile_blocks = 0;
    impop_scan_plan *plan = nullptr;
    int rc = impop_scan_plan_create(ctx, m, windows, n_windows, mask_p, mask_a, mask_b, &sp, &plan);
    if (rc) return rc;
    auto fail = [&](int code) {
        impop_scan_plan_destroy(plan);
        return code;
    };
    // subset P index list and A/B flags
    std::vector<uint32_t> idx;
    std::vector<uint8_t> fa(n, 0), fb(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const bool inP = mask_p ? ((mask_p[i >> 6] >> (i & 63)) & 1ull) : true;
        if (inP) idx.push_back(i);
        fa[i] = mask_a ? (uint8_t)((mask_a[i >> 6] >> (i & 63)) & 1ull) : 0;
        fb[i] = mask_b ? (uint8_t)((mask_b[i >> 6] >> (i & 63)) & 1ull) : 0;
    }
    const uint32_t nP = (uint32_t)idx.size();
    // windows are processed in chunks so the Gram scratch stays bounded (<= ~1 GiB)
    const size_t gram_bytes = (size_t)ld * ld * 4;
    uint64_t chunk = (1ull << 30) / gram_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > 4096) chunk = 4096;
    if (chunk > n_windows) chunk = n_windows;
    void *d = nullptr;
    const size_t need = 4096 + chunk * (gram_bytes + sizeof(GramWindow) + 16 + sizeof(Pica2Out) + sizeof(HfstOut) +
                                        sizeof(impop_window_stats) + sizeof(impop_pairwise_stats) + 2048) +
                        (size_t)n * 8 + 4096;
    rc = ctx_scratch(ctx, need, &d);
    if (rc) return fail(rc);
    Carve2 cv(d);
    int32_t *d_g = cv.take<int32_t>(chunk * (size_t)ld * ld);
    GramWindow *d_w = cv.take<GramWindow>(chunk);
    uint64_t *d_W = cv.take<uint64_t>(chunk);
    uint64_t *d_L = cv.take<uint64_t>(chunk);
    Pica2Out *d_p = cv.take<Pica2Out>(chunk);
    HfstOut *d_h = cv.take<HfstOut>(chunk);
    impop_window_stats *d_s = cv.take<impop_window_stats>(chunk);
    impop_pairwise_stats *d_o = cv.take<impop_pairwise_stats>(chunk);
    uint32_t *d_idx = cv.take<uint32_t>(n ? n : 1);
    uint8_t *d_fa = cv.take<uint8_t>(n ? n : 1);
    uint8_t *d_fb = cv.take<uint8_t>(n ? n : 1);
    hipError_t e;
#define PW_TRY(expr) \
    if ((e = (expr)) != hipSuccess) return fail(hip_fail(e, #expr, __FILE__, __LINE__))
    if (nP) PW_TRY(hipMemcpyAsync(d_idx, idx.data(), (size_t)nP * 4, hipMemcpyHostToDevice, ctx->stream));
    PW_TRY(hipMemcpyAsync(d_fa, fa.data(), n, hipMemcpyHostToDevice, ctx->stream));
    PW_TRY(hipMemcpyAsync(d_fb, fb.data(), n, hipMemcpyHostToDevice, ctx->stream));
    rc = impop_scan_plan_launch(plan, nullptr);
    if (rc) return fail(rc);
    std::vector<impop_window_stats> scan_host(n_windows);
    rc = impop_scan_plan_fetch(plan, scan_host.data());
    if (rc) return fail(rc);
    std::vector<GramWindow> gw(chunk);
    std::vector<uint64_t> Wv(chunk), Lv(chunk);
    for (uint64_t base = 0; base < n_windows; base += chunk) {
        const uint64_t cnt = std::min<uint64_t>(chunk, n_windows - base);
        for (uint64_t k = 0; k < cnt; ++k) {
            gw[k] = {windows[base + k].site_begin, windows[base + k].site_end};
            Wv[k] = windows[base + k].site_end - windows[base + k].site_begin;
            Lv[k] = windows[base + k].seq_len;
        }
        PW_TRY(hipMemcpyAsync(d_w, gw.data(), cnt * sizeof(GramWindow), hipMemcpyHostToDevice, ctx->stream));
        PW_TRY(hipMemcpyAsync(d_W, Wv.data(), cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        PW_TRY(hipMemcpyAsync(d_L, Lv.data(), cnt * 8, hipMemcpyHostToDevice, ctx->stream));
        PW_TRY(hipMemcpyAsync(d_s, scan_host.data() + base, cnt * sizeof(impop_window_stats), hipMemcpyHostToDevice,
                              ctx->stream));
        rc = launch_gram(ctx, m, d_w, (uint32_t)cnt, d_g);
        if (rc) return fail(rc);
        SimBatch b{};
        b.gram = d_g; b.stride = (uint64_t)ld * ld; b.ld = ld; b.n = n; b.W = d_W; b.kind = params->identity_kind;
        b.round_digits = params->round_digits < 0 ? -1 : params->round_digits;
        rc = launch_pica2(ctx, b, cnt, mask_p ? d_idx : nullptr, nP, params->threshold, d_L, d_p, nullptr);
        if (rc) return fail(rc);
        rc = launch_hfst(ctx, b, cnt, d_fa, d_fb, d_L, d_h);
        if (rc) return fail(rc);
        PairFinalIn in{d_p, d_h, d_s};
        hipLaunchKernelGGL(pairwise_finalize_kernel, dim3((uint32_t)((cnt + 63) / 64)), dim3(64), 0, ctx->stream, in, cnt,
                           nP, params->d_pi_mode, params->s_scope, d_o);
        PW_TRY(hipGetLastError());
        PW_TRY(hipMemcpyAsync(out_host + base, d_o, cnt * sizeof(impop_pairwise_stats), hipMemcpyDeviceToHost, ctx->stream));
        PW_TRY(hipStreamSynchronize(ctx->stream));  // gw/Wv/Lv are reused by the next chunk
    }
#undef PW_TRY
    impop_scan_plan_destroy(plan);
    return IMPOP_OK;
}
