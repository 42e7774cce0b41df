// scan.hip — the HBM-bound hot path: one streaming pass over the SB64 presence matrix gives,
// per window, S, sum c(n-c) for the subset P and populations A, B, the A x B cross sum, and
// from those pi, Hudson Fst and Tajima's D.
//
// Work decomposition: windows -> elementary segments (so overlapping / sliding windows read
// every site once) -> tiles of <= tile_blocks 64-site blocks.  One 256-thread workgroup per
// tile; wave w takes blocks b0+w, b0+w+4, ...; lane l of a wave owns site 64b+l and pulls its
// wps dwords with ceil(wps/4) fully coalesced 1 KiB wave loads (layout: internal.h).  The
// three population masks are wave-uniform and live in SGPRs.  Integer partials per tile are
// written once (no atomics => deterministic); a second tiny kernel sums each window's tiles
// and evaluates the fp64 statistics in the reference's operation order.
#include <algorithm>
#include <map>
#include <vector>

#include "device_utils.h"
#include "internal.h"

namespace impop {

struct ScanTile {
    uint64_t site_begin, site_end;
};
struct TilePartial {  // 48 B
    uint32_t s_all, s_p, s_a, s_b;
    uint64_t sum_p, sum_a, sum_b, sum_ab;
};
struct WinDesc {
    uint64_t t0, t1;  // tile range
    uint64_t n_sites;
    uint64_t seq_len;
};
struct PopSizes {
    uint32_t n, nP, nA, nB;
};

template <int WPS>
struct MaskArgs {
    uint32_t p[WPS], a[WPS], b[WPS];
};

// Tuning knobs (defaults chosen by tools/tune_scan.py on MI355X)
// Measured (tools/tune_scan.py, 465 x 75 M sites, interleaved rounds): nt loads 6.65-6.68 TB/s
// algorithmic vs 5.9-6.07 TB/s with default-policy loads (+10 %); unroll / occupancy / tile
// size move the result by < 2 % once nt is on; a 64-VGPR cap (min_waves 8 at unroll 2) spills.
#ifndef IMPOP_SCAN_NT
#define IMPOP_SCAN_NT 1        // 1: non-temporal (streaming) loads for the once-read matrix
#endif
#ifndef IMPOP_SCAN_UNROLL
#define IMPOP_SCAN_UNROLL 0    // 64-site blocks in flight per wave; 0 = auto (about 8 wave loads in flight)
#endif
#ifndef IMPOP_SCAN_MIN_WAVES
#define IMPOP_SCAN_MIN_WAVES 6 // __launch_bounds__ 2nd argument (waves per SIMD)
#endif

template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#if IMPOP_SCAN_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32v2 __attribute__((ext_vector_type(2)));

template <int WPS>
__device__ __forceinline__ void load_site(const uint32_t *__restrict__ blk, uint32_t lane, uint32_t (&w)[WPS]) {
    constexpr int G = (WPS + 3) / 4;
    constexpr int R = WPS - 4 * (G - 1);
#pragma unroll
    for (int g = 0; g < G - 1; ++g) {
        const u32v4 v = stream_load(reinterpret_cast<const u32v4 *>(blk + g * 256 + lane * 4));
        w[4 * g + 0] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w;
    }
    const uint32_t *last = blk + (G - 1) * 256 + lane * R;
    if constexpr (R == 4) {
        const u32v4 v = stream_load(reinterpret_cast<const u32v4 *>(last));
        w[4 * (G - 1) + 0] = v.x; w[4 * (G - 1) + 1] = v.y; w[4 * (G - 1) + 2] = v.z; w[4 * (G - 1) + 3] = v.w;
    } else if constexpr (R == 3) {
        // 12-byte, 4-byte-aligned: three dwords (the backend merges them into one dwordx3)
        w[4 * (G - 1) + 0] = stream_load(last);
        w[4 * (G - 1) + 1] = stream_load(last + 1);
        w[4 * (G - 1) + 2] = stream_load(last + 2);
    } else if constexpr (R == 2) {
        const u32v2 v = stream_load(reinterpret_cast<const u32v2 *>(last));
        w[4 * (G - 1) + 0] = v.x; w[4 * (G - 1) + 1] = v.y;
    } else {
        w[4 * (G - 1)] = stream_load(last);
    }
}

struct LaneAcc {
    uint32_t s_all = 0, s_p = 0, s_a = 0, s_b = 0;
    uint64_t q_p = 0, q_a = 0, q_b = 0, q_ab = 0;
};

// Fixed-WPS kernel (n <= 512): every product c (n - c) is below 2^16, so the lane accumulators of one
// tile fit 32 bits (<= 1024 sites per lane and tile) and the multiply is a 24-bit v_mul / v_mad
// (v_mul_lo_u32 is a quarter-rate instruction); they are widened once, at the tile reduction.  This is
// what lifts the small-n cases, which are VALU-bound rather than HBM-bound (4-16 B per site).
struct LaneAcc32 {
    uint32_t s_all = 0, s_p = 0, s_a = 0, s_b = 0;
    uint32_t q_p = 0, q_a = 0, q_b = 0, q_ab = 0;
};

// One site per lane.  No validity test here: lanes outside the tile had their words zeroed by the caller
// (only the first / last block of a tile can be partial), and all-zero words add nothing to any counter.
// Segregating test 0 < c < n as ONE unsigned compare: (c - 1) < (n - 1)  (false for n = 0 or 1 as well).
template <int WPS, bool SUBSET_P>
__device__ __forceinline__ void site_accumulate(const uint32_t (&w)[WPS], const MaskArgs<WPS> &mk, const PopSizes &ps,
                                                LaneAcc32 &acc) {
    uint32_t c = 0, cP = 0, cA = 0, cB = 0;
#pragma unroll
    for (int k = 0; k < WPS; ++k) {
        c += __popc(w[k]);
        if (SUBSET_P) cP += __popc(w[k] & mk.p[k]);
        cA += __popc(w[k] & mk.a[k]);
        cB += __popc(w[k] & mk.b[k]);
    }
    if (!SUBSET_P) cP = c;
    acc.s_all += (c - 1u) < (ps.n - 1u);
    acc.s_p += (cP - 1u) < (ps.nP - 1u);
    acc.s_a += (cA - 1u) < (ps.nA - 1u);
    acc.s_b += (cB - 1u) < (ps.nB - 1u);
    const uint32_t rA = ps.nA - cA, rB = ps.nB - cB;
    acc.q_p += __umul24(cP, ps.nP - cP);
    acc.q_a += __umul24(cA, rA);
    acc.q_b += __umul24(cB, rB);
    acc.q_ab += __umul24(cA, rB) + __umul24(cB, rA);
}

__device__ __forceinline__ void tile_reduce_store(LaneAcc &acc, TilePartial *out) {
    __shared__ uint64_t red[4][8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t s0 = wave_sum_u32(acc.s_all), s1 = wave_sum_u32(acc.s_p), s2 = wave_sum_u32(acc.s_a),
                   s3 = wave_sum_u32(acc.s_b);
    const uint64_t q0 = wave_sum_u64(acc.q_p), q1 = wave_sum_u64(acc.q_a), q2 = wave_sum_u64(acc.q_b),
                   q3 = wave_sum_u64(acc.q_ab);
    if (lane == 0) {
        red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3;
        red[wave][4] = q0; red[wave][5] = q1; red[wave][6] = q2; red[wave][7] = q3;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const uint64_t v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        TilePartial *o = out + blockIdx.x;
        switch (threadIdx.x) {
            case 0: o->s_all = (uint32_t)v; break;
            case 1: o->s_p = (uint32_t)v; break;
            case 2: o->s_a = (uint32_t)v; break;
            case 3: o->s_b = (uint32_t)v; break;
            case 4: o->sum_p = v; break;
            case 5: o->sum_a = v; break;
            case 6: o->sum_b = v; break;
            default: o->sum_ab = v; break;
        }
    }
}

template <int WPS, bool SUBSET_P>
__global__ __launch_bounds__(256, IMPOP_SCAN_MIN_WAVES) void scan_tiles_kernel(const uint32_t *__restrict__ sb,
                                                                               const ScanTile *__restrict__ tiles,
                                                                               const MaskArgs<WPS> mk, const PopSizes ps,
                                                                               TilePartial *__restrict__ out) {
    const ScanTile t = tiles[blockIdx.x];
    const uint64_t b0 = t.site_begin >> 6, b1 = (t.site_end + 63) >> 6;
    // wave index through readfirstlane: block addresses and the loop stay scalar (SGPR) state
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LaneAcc32 acc;
    uint64_t b = b0 + wave;
    constexpr int G = (WPS + 3) / 4;
    constexpr int U = IMPOP_SCAN_UNROLL > 0 ? IMPOP_SCAN_UNROLL : (G >= 3 ? 2 : G == 2 ? 4 : 8);
    // Only the first and the last block of a tile can be partial: lanes before `lo` in block b0 and from `hi`
    // on in block b1-1 get all-zero words (wave-uniform branch, taken for those two blocks only); interior
    // blocks carry no validity arithmetic and no branch, so all U loads of an iteration are in flight together.
    const uint32_t lo = (uint32_t)(t.site_begin - (b0 << 6));
    const uint32_t hi = (uint32_t)(t.site_end - ((b1 - 1) << 6));  // 1..64
    auto trim = [&](uint32_t (&w)[WPS], uint64_t blk) {
        if ((blk == b0 && lo != 0) | (blk == b1 - 1 && hi != 64)) {
            const bool keep = lane >= (blk == b0 ? lo : 0u) && lane < (blk == b1 - 1 ? hi : 64u);
#pragma unroll
            for (int k = 0; k < WPS; ++k) w[k] = keep ? w[k] : 0u;
        }
    };
    // U blocks per iteration: U*ceil(WPS/4) independent 1 KiB wave loads in flight per wave
    for (; b + 4 * (U - 1) < b1; b += 4 * U) {
        uint32_t w[U][WPS];
#pragma unroll
        for (int u = 0; u < U; ++u) load_site<WPS>(sb + (b + 4 * u) * (64ull * WPS), lane, w[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            trim(w[u], b + 4 * u);
            site_accumulate<WPS, SUBSET_P>(w[u], mk, ps, acc);
        }
    }
    for (; b < b1; b += 4) {
        uint32_t w0[WPS];
        load_site<WPS>(sb + b * (64ull * WPS), lane, w0);
        trim(w0, b);
        site_accumulate<WPS, SUBSET_P>(w0, mk, ps, acc);
    }
    LaneAcc wide;
    wide.s_all = acc.s_all; wide.s_p = acc.s_p; wide.s_a = acc.s_a; wide.s_b = acc.s_b;
    wide.q_p = acc.q_p; wide.q_a = acc.q_a; wide.q_b = acc.q_b; wide.q_ab = acc.q_ab;
    tile_reduce_store(wide, out);
}

// Any wps (n > 512 haplotypes, and every weighted matrix): the three masks of the WHOLE haplotype axis sit in
// LDS (3 x wps dwords, <= 24 KB at the 65 535-haplotype limit; filled once per workgroup) and come back as
// broadcast ds_read_b128 — every lane the same address, no bank conflict — so a wave simply streams the
// granules of one 64-site block after the other, AN_U fully coalesced 1 KiB loads in flight, with the four
// per-site counts in registers for the length of the block.  Every byte of the tile is read exactly once.
// (Round 1 walked the haplotype axis in 16-dword chunks whose masks were re-loaded into SGPRs per chunk for up
// to 8 blocks per wave: 6.08 TB/s on the 4096 x 10^7 launch, and tiles were pinned to 32 blocks = 1 MB, which
// left a fifth of the chip idle in the last wave of tiles.)
// WEIGHTED (impop_matrix_set_site_weights): column s stands for w_s base pairs (a graph node of that length), so
// every sum_s c (n - c) becomes sum_s w_s c (n - c) and the window's W is sum_s w_s (host prefix sums at plan
// time) — exactly what scanning the bp-expanded matrix gives — while the segregating-site counts stay counts
// of COLUMNS (variable nodes, what a VCF of the window lists).
constexpr int AN_U = 8;  // granules (1 KiB wave loads) in flight per wave

template <bool SUBSET_P>
__device__ __forceinline__ void anyn_granule(const u32v4 v, const uint32_t *lp, const uint32_t *la, const uint32_t *lb,
                                             uint32_t g, uint32_t &c, uint32_t &cP, uint32_t &cA, uint32_t &cB) {
    const u32v4 ka = *reinterpret_cast<const u32v4 *>(la + 4 * g);  // broadcast LDS reads
    const u32v4 kb = *reinterpret_cast<const u32v4 *>(lb + 4 * g);
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    cA += __popc(v.x & ka.x) + __popc(v.y & ka.y) + __popc(v.z & ka.z) + __popc(v.w & ka.w);
    cB += __popc(v.x & kb.x) + __popc(v.y & kb.y) + __popc(v.z & kb.z) + __popc(v.w & kb.w);
    if (SUBSET_P) {
        const u32v4 kp = *reinterpret_cast<const u32v4 *>(lp + 4 * g);
        cP += __popc(v.x & kp.x) + __popc(v.y & kp.y) + __popc(v.z & kp.z) + __popc(v.w & kp.w);
    }
}

template <bool SUBSET_P, bool WEIGHTED>
__global__ __launch_bounds__(256, 4) void scan_tiles_anyn_kernel(const uint32_t *__restrict__ sb, const ScanTile *__restrict__ tiles,
                                                                 const uint32_t *__restrict__ masks, uint32_t wps, uint32_t G,
                                                                 uint32_t r, const PopSizes ps, const uint32_t *__restrict__ weights,
                                                                 TilePartial *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t an_lds[];  // p | a | b, each padded to a multiple of 4 dwords
    const uint32_t wps4 = (wps + 3) & ~3u;
    uint32_t *lp = an_lds, *la = an_lds + wps4, *lb = an_lds + 2 * wps4;
    for (uint32_t k = threadIdx.x; k < wps4; k += 256) {
        const bool in = k < wps;
        lp[k] = in ? masks[k] : 0u; la[k] = in ? masks[wps + k] : 0u; lb[k] = in ? masks[2 * wps + k] : 0u;
    }
    __syncthreads();
    const ScanTile t = tiles[blockIdx.x];
    const uint64_t b0 = t.site_begin >> 6, b1 = (t.site_end + 63) >> 6;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // full 16-byte granules: all of them when the last one holds 4 dwords per site (r == 4: its addressing is
    // the full granules'), else all but the last, whose r = 1..3 dwords per site are read one by one
    const uint32_t Gf = r == 4 ? G : G - 1;
    LaneAcc acc;
    for (uint64_t b = b0 + wave; b < b1; b += 4) {
        const uint32_t *blk = sb + b * 64ull * wps + lane * 4;
        uint32_t c = 0, cP = 0, cA = 0, cB = 0;
        // the last granule's r = 1..3 dwords per site go out first, together with the first batch (read behind the
        // batches they cost one more memory latency per block)
        uint32_t tail[3] = {0u, 0u, 0u};
        if (Gf < G) {
            const uint32_t *last = sb + b * 64ull * wps + (uint64_t)Gf * 256 + lane * r;
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if ((uint32_t)e < r) tail[e] = stream_load(last + e);
        }
        // full batches of AN_U granules (all loads out first, consumed with staggered waits) ...
        uint32_t g = 0;
        for (; g + AN_U <= Gf; g += AN_U) {
            u32v4 v[AN_U];
#pragma unroll
            for (int u = 0; u < AN_U; ++u) v[u] = stream_load(reinterpret_cast<const u32v4 *>(blk + (uint64_t)(g + u) * 256));
#pragma unroll
            for (int u = 0; u < AN_U; ++u) anyn_granule<SUBSET_P>(v[u], lp, la, lb, g + u, c, cP, cA, cB);
        }
        // ... and ONE short batch for what is left (wave-uniform predicates): its loads still go out together; a
        // one-at-a-time remainder loop paid the full HBM latency per granule
        if (g < Gf) {
            const uint32_t nb = Gf - g;
            u32v4 v[AN_U];
#pragma unroll
            for (int u = 0; u < AN_U - 1; ++u)
                if ((uint32_t)u < nb) v[u] = stream_load(reinterpret_cast<const u32v4 *>(blk + (uint64_t)(g + u) * 256));
#pragma unroll
            for (int u = 0; u < AN_U - 1; ++u)
                if ((uint32_t)u < nb) anyn_granule<SUBSET_P>(v[u], lp, la, lb, g + u, c, cP, cA, cB);
        }
        if (Gf < G) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if ((uint32_t)e < r) {
                    const uint32_t v = tail[e], k = 4 * Gf + e;
                    c += __popc(v); cA += __popc(v & la[k]); cB += __popc(v & lb[k]);
                    if (SUBSET_P) cP += __popc(v & lp[k]);
                }
        }
        if (!SUBSET_P) cP = c;
        const uint64_t s = b * 64 + lane;
        if (s >= t.site_begin && s < t.site_end) {  // only the first / last block of a tile is partial
            acc.s_all += (c - 1u) < (ps.n - 1u);
            acc.s_p += (cP - 1u) < (ps.nP - 1u);
            acc.s_a += (cA - 1u) < (ps.nA - 1u);
            acc.s_b += (cB - 1u) < (ps.nB - 1u);
            // n <= 65535: every product is below 2^32
            const uint32_t qp = cP * (ps.nP - cP), qa = cA * (ps.nA - cA), qb = cB * (ps.nB - cB);
            const uint64_t qab = (uint64_t)(cA * (ps.nB - cB)) + (uint64_t)(cB * (ps.nA - cA));
            if (WEIGHTED) {
                const uint64_t wt = weights[s];
                acc.q_p += wt * qp; acc.q_a += wt * qa; acc.q_b += wt * qb; acc.q_ab += wt * qab;
            } else {
                acc.q_p += qp; acc.q_a += qa; acc.q_b += qb; acc.q_ab += qab;
            }
        }
    }
    tile_reduce_store(acc, out);
}

// TPW threads per window sum its tile partials (integers: any order gives the same totals), then one thread
// runs the fp64 epilogue.  Same operation order as oracle_window_sitecount (oracle/impop_oracle.c) which restates
// pica2.py:154,164, h-fst.py:203-240 and tj_d.py:53-65 on the exact pair sums.  TPW = 1 for the usual many-
// windows-of-a-few-tiles shape, 64 (a wave per window) / 256 (a workgroup per window) when windows span many
// tiles — BASELINE config 5's single 10^7-site window is ~19 500 tiles, a 0.9 MB read that one thread would
// walk serially behind the streaming kernel.
struct WinTotals {
    uint32_t s_all, s_p, s_a, s_b;
    uint64_t sum_p, sum_a, sum_b, sum_ab;
};

__device__ inline void window_epilogue(const WinDesc &w, const WinTotals &T, const PopSizes &ps, const double *__restrict__ taj,
                                       int d_pi_mode, int s_scope, impop_window_stats *__restrict__ dst) {
    const uint64_t n_sites = w.n_sites;  // window length, or the sum of its columns' weights (window_weights)
    impop_window_stats r;
    r.n_sites = (uint32_t)n_sites;       // <= 2^32 - 1 by the plan-time checks
    r.s_all = T.s_all; r.s_p = T.s_p; r.s_a = T.s_a; r.s_b = T.s_b; r.flags = 0;
    r.sum_p = T.sum_p; r.sum_a = T.sum_a; r.sum_b = T.sum_b; r.sum_ab = T.sum_ab;
    const double nan = __builtin_nan("");
    const double W = (double)n_sites;
    const double seq_len = (double)w.seq_len;
    const double nP = (double)ps.nP, nA = (double)ps.nA, nB = (double)ps.nB;
    const double pairsP = nP * (double)(ps.nP - 1) / 2.0;
    const double pi = (ps.nP >= 2 && W > 0) ? (double)T.sum_p / (pairsP * W) : 0.0;
    const double pi_site = (seq_len != 0.0) ? pi / seq_len : nan;
    const double pairsA = nA * (nA - 1.0) / 2.0, pairsB = nB * (nB - 1.0) / 2.0;
    double pi_a = (ps.nA >= 2 && W > 0) ? (double)T.sum_a / (pairsA * W) : 0.0;
    double pi_b = (ps.nB >= 2 && W > 0) ? (double)T.sum_b / (pairsB * W) : 0.0;
    double dxy = (ps.nA && ps.nB && W > 0) ? (double)T.sum_ab / (nA * nB * W) : 0.0;
    double pi_xy = 0.5 * (pi_a + pi_b);
    const double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;
    double da = dxy - pi_xy;
    if (seq_len > 0) {
        pi_a /= seq_len; pi_b /= seq_len; da = (dxy - pi_xy) / seq_len; pi_xy /= seq_len; dxy /= seq_len;
    }
    r.pi = pi; r.pi_site = pi_site; r.pi_a = pi_a; r.pi_b = pi_b; r.pi_xy = pi_xy; r.dxy = dxy; r.da = da; r.fst = fst;
    const double S = (double)(s_scope == 0 ? T.s_all : T.s_p);
    const double pin = d_pi_mode == 0 ? py_round(pi_site, 8) : d_pi_mode == 1 ? pi_site : pi * W;
    double D = nan;
    if (ps.nP >= 2 && pin == pin) {
        TajConsts c;
        c.a1 = taj[0]; c.a2 = taj[1]; c.b1 = taj[2]; c.b2 = taj[3]; c.c1 = taj[4]; c.c2 = taj[5]; c.e1 = taj[6]; c.e2 = taj[7];
        D = tajima_d_from(c, S, pin, nullptr, nullptr);
    }
    r.tajima_d = D;
    *dst = r;
}

template <int TPW>
__global__ __launch_bounds__(TPW == 1 ? 128 : 256) void scan_finalize_kernel(const TilePartial *__restrict__ parts,
                                                                             const WinDesc *__restrict__ wins, uint64_t n_windows,
                                                                             PopSizes ps, const double *__restrict__ taj,
                                                                             int d_pi_mode, int s_scope,
                                                                             impop_window_stats *__restrict__ out) {
    constexpr int BLOCK = TPW == 1 ? 128 : 256;
    const uint64_t i = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) / TPW;  // uniform per wave for TPW >= 64
    const uint32_t sub = threadIdx.x % TPW;
    if (TPW < 256 && i >= n_windows) return;
    const WinDesc w = wins[i];
    WinTotals T = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t t = w.t0 + sub; t < w.t1; t += TPW) {
        const TilePartial p = parts[t];
        T.s_all += p.s_all; T.s_p += p.s_p; T.s_a += p.s_a; T.s_b += p.s_b;
        T.sum_p += p.sum_p; T.sum_a += p.sum_a; T.sum_b += p.sum_b; T.sum_ab += p.sum_ab;
    }
    if (TPW >= 64) {
        T.s_all = wave_sum_u32(T.s_all); T.s_p = wave_sum_u32(T.s_p); T.s_a = wave_sum_u32(T.s_a); T.s_b = wave_sum_u32(T.s_b);
        T.sum_p = wave_sum_u64(T.sum_p); T.sum_a = wave_sum_u64(T.sum_a); T.sum_b = wave_sum_u64(T.sum_b);
        T.sum_ab = wave_sum_u64(T.sum_ab);
    }
    if (TPW == 256) {
        __shared__ WinTotals red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = T;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 1; k < 4; ++k) {
                T.s_all += red[k].s_all; T.s_p += red[k].s_p; T.s_a += red[k].s_a; T.s_b += red[k].s_b;
                T.sum_p += red[k].sum_p; T.sum_a += red[k].sum_a; T.sum_b += red[k].sum_b; T.sum_ab += red[k].sum_ab;
            }
    }
    if (sub == 0) window_epilogue(w, T, ps, taj, d_pi_mode, s_scope, out + i);
}

// c = sum_k popc(dword_k & mask_k) of site `lane` of a block: whole 16-byte granules (one coalesced 1 KiB wave load
// each, all of them and the last granule's dwords issued before the first is consumed when G <= 5), masks wave-uniform
__device__ __forceinline__ uint32_t masked_site_count(const uint32_t *__restrict__ blk, const uint32_t *__restrict__ mask, uint32_t G,
                                                      uint32_t r, uint32_t lane) {
    uint32_t c = 0;
    const uint32_t Gf = r == 4 ? G : G - 1;
    const uint32_t *last = blk + (uint64_t)Gf * 256 + lane * r;
    uint32_t tl[3] = {0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (Gf < G && (uint32_t)j < r) tl[j] = stream_load(last + j);
    for (uint32_t g = 0; g < Gf; g += 4) {
        const uint32_t nb = Gf - g < 4u ? Gf - g : 4u;
        u32v4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((uint32_t)u < nb) v[u] = stream_load(reinterpret_cast<const u32v4 *>(blk + (uint64_t)(g + u) * 256 + lane * 4));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((uint32_t)u < nb) {
                const uint32_t *m = mask + 4 * (g + u);
                c += __popc(v[u].x & m[0]) + __popc(v[u].y & m[1]) + __popc(v[u].z & m[2]) + __popc(v[u].w & m[3]);
            }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (Gf < G && (uint32_t)j < r) c += __popc(tl[j] & mask[4 * Gf + j]);
    return c;
}

__global__ __launch_bounds__(256) void site_counts_kernel(const uint32_t *__restrict__ sb, const uint32_t *__restrict__ mask,
                                                          uint32_t wps, uint32_t G, uint32_t r, uint64_t site_begin,
                                                          uint64_t site_end, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = (site_begin >> 6) + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t s = b * 64 + lane;
    if (s < site_begin || s >= site_end) return;
    out[s - site_begin] = masked_site_count(sb + b * 64ull * wps, mask, G, r, lane);
}

// ---- K disjoint populations in one pass: all-pairs Hudson Fst (run_h_fst_panels.sh:60-71) -------
// Per tile: sum_k c_k (n_k - c_k) for every population and sum_s [c_k (n_l - c_l) + c_l (n_k - c_k)] for
// every pair k < l.  Masks come from memory (wave-uniform scalar loads); K is a template parameter so
// that the per-lane accumulators are registers.  out: tile-major, K + K(K-1)/2 uint64 per tile.
// Round 2: masks of all K populations in LDS (K x wps dwords, broadcast ds_read_b128), granules loaded four at a time
// (the first version had ONE load in flight per wave), and for unweighted matrices of <= 512 haplotypes (SMALL) the
// per-site work is K + K + K(K-1)/2 32-bit multiply-adds — sum c_k, sum c_k^2, sum c_k c_l — from which the tile's
//   sum_s c_k (n_k - c_k) = n_k sum c_k - sum c_k^2,   sum_s [c_k (n_l - c_l) + c_l (n_k - c_k)] = n_l sum c_k + n_k sum c_l - 2 sum c_k c_l
// follow once per lane and tile (every partial sum stays below 2^32: <= 1024 sites per lane and tile, products < 2^18);
// the first version did three to four 64-bit multiplies per pair and site, which made K = 8 VALU-bound.
template <int K, bool SMALL>
__global__ __launch_bounds__(256) void scan_multi_kernel(const uint32_t *__restrict__ sb, const ScanTile *__restrict__ tiles,
                                                         const uint32_t *__restrict__ masks /* K x wps */,
                                                         const uint32_t *__restrict__ pop_n /* K */, uint32_t wps,
                                                         uint32_t G, uint32_t r, const uint32_t *__restrict__ weights /* nullable */,
                                                         uint64_t *__restrict__ out) {
    constexpr int NP = K * (K - 1) / 2;
    constexpr int MU = 4;  // granules in flight per wave
    extern __shared__ __attribute__((aligned(16))) uint32_t mk_lds[];  // K x wps4
    __shared__ uint64_t red[4][K + NP];
    const uint32_t wps4 = (wps + 3) & ~3u;
    for (uint32_t i = threadIdx.x; i < K * wps4; i += 256) {
        const uint32_t k = i / wps4, j = i % wps4;
        mk_lds[i] = j < wps ? masks[(uint64_t)k * wps + j] : 0u;
    }
    __syncthreads();
    const ScanTile t = tiles[blockIdx.x];
    const uint64_t b0 = t.site_begin >> 6, b1 = (t.site_end + 63) >> 6;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t nk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) nk[k] = pop_n[k];
    uint64_t acc[K + NP];
    uint32_t s1[K], q2[K], px[NP > 0 ? NP : 1];
#pragma unroll
    for (int i = 0; i < K + NP; ++i) acc[i] = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) { s1[k] = 0; q2[k] = 0; }
#pragma unroll
    for (int i = 0; i < NP; ++i) px[i] = 0;
    const uint32_t Gf = r == 4 ? G : G - 1;  // full 16-byte granules (see scan_tiles_anyn_kernel)
    auto load_batch = [&](const uint32_t *blk, uint32_t g, uint32_t nb, u32v4 (&v)[MU]) {
#pragma unroll
        for (int u = 0; u < MU; ++u)
            if ((uint32_t)u < nb) v[u] = stream_load(reinterpret_cast<const u32v4 *>(blk + (uint64_t)(g + u) * 256 + lane * 4));
    };
    auto count_batch = [&](uint32_t g, uint32_t nb, const u32v4 (&v)[MU], uint32_t (&c)[K]) {
#pragma unroll
        for (int u = 0; u < MU; ++u)
            if ((uint32_t)u < nb) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const u32v4 m4 = *reinterpret_cast<const u32v4 *>(mk_lds + k * wps4 + 4 * (g + u));
                    c[k] += __popc(v[u].x & m4.x) + __popc(v[u].y & m4.y) + __popc(v[u].z & m4.z) + __popc(v[u].w & m4.w);
                }
            }
    };
    // the last granule's r = 1..3 dwords per site (r == 4 counts as a full granule): loaded TOGETHER with the batch —
    // read one by one behind it they cost three more memory latencies per block
    auto load_tail = [&](const uint32_t *blk, uint32_t (&tl)[3]) {
        const uint32_t *last = blk + (uint64_t)Gf * 256 + lane * r;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (Gf < G && (uint32_t)j < r) tl[j] = stream_load(last + j);
    };
    auto count_tail = [&](const uint32_t (&tl)[3], uint32_t (&c)[K]) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (Gf < G && (uint32_t)j < r) {
#pragma unroll
                for (int k = 0; k < K; ++k) c[k] += __popc(tl[j] & mk_lds[k * wps4 + 4 * Gf + j]);
            }
    };
    auto tally = [&](uint64_t b, const uint32_t (&c)[K]) {
        const uint64_t s = b * 64 + lane;
        if (s >= t.site_begin && s < t.site_end) {
            if (SMALL) {
                int pi = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    s1[k] += c[k];
                    q2[k] += __umul24(c[k], c[k]);
#pragma unroll
                    for (int l = k + 1; l < K; ++l) px[pi++] += __umul24(c[k], c[l]);
                }
            } else {
                const uint64_t wt = weights ? weights[s] : 1;  // wave-uniform choice
                int pi = K;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    acc[k] += wt * (c[k] * (nk[k] - c[k]));
#pragma unroll
                    for (int l = k + 1; l < K; ++l) acc[pi++] += wt * (c[k] * (nk[l] - c[l]) + c[l] * (nk[k] - c[k]));
                }
            }
        }
    };
    uint64_t b = b0 + wave;
    if (Gf <= (uint32_t)MU) {
        // <= 512 haplotypes: a block is one batch; two blocks (up to 8 wave loads) in flight per wave
        for (; b + 4 < b1; b += 8) {
            const uint32_t *blk0 = sb + b * 64ull * wps, *blk1 = sb + (b + 4) * 64ull * wps;
            u32v4 v0[MU], v1[MU];
            uint32_t t0[3], t1[3];
            load_batch(blk0, 0, Gf, v0);
            load_tail(blk0, t0);
            load_batch(blk1, 0, Gf, v1);
            load_tail(blk1, t1);
            uint32_t c0[K], c1[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { c0[k] = 0; c1[k] = 0; }
            count_batch(0, Gf, v0, c0);
            count_tail(t0, c0);
            tally(b, c0);
            count_batch(0, Gf, v1, c1);
            count_tail(t1, c1);
            tally(b + 4, c1);
        }
    }
    for (; b < b1; b += 4) {
        const uint32_t *blk = sb + b * 64ull * wps;
        uint32_t c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = 0;
        uint32_t tl[3];
        load_tail(blk, tl);
        for (uint32_t g = 0; g < Gf; g += MU) {
            const uint32_t nb = Gf - g < (uint32_t)MU ? Gf - g : (uint32_t)MU;
            u32v4 v[MU];
            load_batch(blk, g, nb, v);
            count_batch(g, nb, v, c);
        }
        count_tail(tl, c);
        tally(b, c);
    }
    if (SMALL) {
        int pi = K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[k] = (uint64_t)nk[k] * s1[k] - q2[k];
#pragma unroll
            for (int l = k + 1; l < K; ++l) { acc[pi] = (uint64_t)nk[l] * s1[k] + (uint64_t)nk[k] * s1[l] - 2ull * px[pi - K]; ++pi; }
        }
    }
#pragma unroll
    for (int i = 0; i < K + NP; ++i) {
        const uint64_t v = wave_sum_u64(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < K + NP)
        out[(uint64_t)blockIdx.x * (K + NP) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one thread per (window, pair): h-fst.py:203-240 on the exact pair sums
__global__ void scan_multi_finalize_kernel(const uint64_t *__restrict__ parts, const WinDesc *__restrict__ wins,
                                           uint64_t n_windows, uint32_t K, const uint32_t *__restrict__ pop_n,
                                           impop_pair_stats *__restrict__ out) {
    const uint32_t NP = K * (K - 1) / 2, stride = K + NP;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_windows * NP) return;
    const uint64_t win = i / NP;
    uint32_t p = (uint32_t)(i % NP), k = 0;
    while (p >= K - 1 - k) { p -= K - 1 - k; ++k; }
    const uint32_t l = k + 1 + p;
    const WinDesc w = wins[win];
    uint64_t sk = 0, sl = 0, skl = 0;
    for (uint64_t t = w.t0; t < w.t1; ++t) {
        sk += parts[t * stride + k];
        sl += parts[t * stride + l];
        skl += parts[t * stride + K + (uint32_t)(i % NP)];
    }
    const double W = (double)w.n_sites, seq_len = (double)w.seq_len;
    const double nA = (double)pop_n[k], nB = (double)pop_n[l];
    const double pairsA = nA * (nA - 1.0) / 2.0, pairsB = nB * (nB - 1.0) / 2.0;
    double pi_a = (pop_n[k] >= 2 && W > 0) ? (double)sk / (pairsA * W) : 0.0;
    double pi_b = (pop_n[l] >= 2 && W > 0) ? (double)sl / (pairsB * W) : 0.0;
    double dxy = (pop_n[k] && pop_n[l] && W > 0) ? (double)skl / (nA * nB * W) : 0.0;
    double pi_xy = 0.5 * (pi_a + pi_b);
    const double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;
    double da = dxy - pi_xy;
    if (seq_len > 0) { pi_a /= seq_len; pi_b /= seq_len; da = (dxy - pi_xy) / seq_len; pi_xy /= seq_len; dxy /= seq_len; }
    impop_pair_stats r;
    r.fst = fst; r.pi_a = pi_a; r.pi_b = pi_b; r.pi_xy = pi_xy; r.dxy = dxy; r.da = da;
    out[i] = r;
}

// ---- allele-frequency spectrum (scripts/wip/op-afs.py): per window, how many sites carry c copies ----
// grid (chunks of 4096 sites, windows); LDS histogram per workgroup, integer atomics to the output.
__global__ __launch_bounds__(256) void afs_kernel(const uint32_t *__restrict__ sb, const uint32_t *__restrict__ mask,
                                                  uint32_t wps, uint32_t G, uint32_t r, const impop_window *__restrict__ wins,
                                                  uint32_t bins, uint64_t chunk_sites, uint32_t *__restrict__ out) {
    extern __shared__ uint32_t hist[];
    const impop_window w = wins[blockIdx.y];
    const uint64_t c0 = w.site_begin + (uint64_t)blockIdx.x * chunk_sites;
    if (c0 >= w.site_end) return;  // uniform per workgroup
    const uint64_t c1 = c0 + chunk_sites < w.site_end ? c0 + chunk_sites : w.site_end;
    for (uint32_t i = threadIdx.x; i < bins; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Most sites of a real matrix are carried by nobody or by everybody: those two bins are counted per WAVE (a ballot
    // and a popcount, one LDS add by one lane) — 64 lanes adding to the same LDS word one after the other was what this
    // kernel spent its time on; the bins in between keep their per-lane atomics (few lanes, spread over many words).
    uint32_t n_zero = 0, n_full = 0;
    for (uint64_t b = (c0 >> 6) + wave; b <= ((c1 - 1) >> 6); b += 4) {
        const uint64_t s = b * 64 + lane;
        const bool in = s >= c0 && s < c1;
        const uint32_t c = masked_site_count(sb + b * 64ull * wps, mask, G, r, lane);
        const bool zero = in && c == 0, full = in && !zero && c == bins - 1;  // an empty mask has ONE bin: count it once
        n_zero += (uint32_t)__popcll(__ballot(zero));
        n_full += (uint32_t)__popcll(__ballot(full));
        if (in && !zero && !full) atomicAdd(&hist[c], 1u);
    }
    if (lane == 0) {
        if (n_zero) atomicAdd(&hist[0], n_zero);
        if (n_full) atomicAdd(&hist[bins - 1], n_full);
    }
    __syncthreads();
    uint32_t *o = out + (uint64_t)blockIdx.y * bins;
    if (gridDim.x == 1) {  // the window's only workgroup: its histogram IS the result
        for (uint32_t i = threadIdx.x; i < bins; i += 256) o[i] = hist[i];
    } else {
        for (uint32_t i = threadIdx.x; i < bins; i += 256)
            if (hist[i]) atomicAdd(&o[i], hist[i]);
    }
}

// mask bitset (uint64 words, n bits) -> wps dwords clipped to n; NULL -> `fill`
static void mask_to_dwords(const uint64_t *mask, uint32_t n, uint32_t wps, bool fill_all, std::vector<uint32_t> &out) {
    out.assign(wps, 0u);
    for (uint32_t k = 0; k < wps; ++k) {
        uint32_t v;
        if (mask) v = (uint32_t)(mask[k >> 1] >> (32 * (k & 1)));
        else v = fill_all ? 0xFFFFFFFFu : 0u;
        const uint32_t lo = 32 * k;
        if (lo + 32 > n) v &= (n > lo) ? (uint32_t)((1ull << (n - lo)) - 1ull) : 0u;
        out[k] = v;
    }
}
static uint32_t popcount_vec(const std::vector<uint32_t> &v) {
    uint32_t c = 0;
    for (uint32_t x : v) c += (uint32_t)__builtin_popcount(x);
    return c;
}

// windows -> elementary segments between sorted window boundaries (a segment is tiled iff some
// window covers it, and exactly once however many windows overlap it) -> tiles of <= tile_blocks
// 64-site blocks; every window becomes a contiguous tile range [t0, t1).
static void build_tiles(const impop_window *windows, uint64_t n_windows, uint32_t tile_blocks, uint32_t wps,
                        std::vector<ScanTile> &tiles, std::vector<WinDesc> &wd, uint64_t &bytes_streamed) {
    std::vector<uint64_t> cuts;
    cuts.reserve(2 * n_windows);
    for (uint64_t i = 0; i < n_windows; ++i)
        if (windows[i].site_end > windows[i].site_begin) {
            cuts.push_back(windows[i].site_begin);
            cuts.push_back(windows[i].site_end);
        }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<int64_t> cover(cuts.size() + 1, 0);
    auto cut_index = [&](uint64_t s) { return (size_t)(std::lower_bound(cuts.begin(), cuts.end(), s) - cuts.begin()); };
    for (uint64_t i = 0; i < n_windows; ++i)
        if (windows[i].site_end > windows[i].site_begin) {
            cover[cut_index(windows[i].site_begin)] += 1;
            cover[cut_index(windows[i].site_end)] -= 1;
        }
    std::vector<uint64_t> seg_tile_start(cuts.size() + 1, 0);
    int64_t depth = 0;
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        seg_tile_start[k] = tiles.size();
        depth += cover[k];
        if (depth <= 0) continue;
        // tiles are cut on 64-site block boundaries of the matrix so interior tiles read whole blocks
        uint64_t s = cuts[k];
        const uint64_t e = cuts[k + 1];
        // equal shares: a 781-block segment under a 512-block limit becomes 391 + 390 blocks, not 512 + 269
        const uint64_t nblk = (e + 63) / 64 - s / 64, n_parts = (nblk + tile_blocks - 1) / tile_blocks;
        const uint64_t per = (nblk + n_parts - 1) / n_parts;
        while (s < e) {
            uint64_t t_end = ((s / 64) + per) * 64;  // block-aligned end
            if (t_end > e) t_end = e;
            tiles.push_back({s, t_end});
            bytes_streamed += ((t_end + 63) / 64 - s / 64) * 64ull * wps * 4ull;
            s = t_end;
        }
    }
    if (!cuts.empty()) seg_tile_start[cuts.size() - 1] = tiles.size();
    seg_tile_start[cuts.size()] = tiles.size();
    wd.resize(n_windows);
    for (uint64_t i = 0; i < n_windows; ++i) {
        WinDesc &w = wd[i];
        w.n_sites = windows[i].site_end - windows[i].site_begin;
        w.seq_len = windows[i].seq_len;
        if (w.n_sites) {
            w.t0 = seg_tile_start[cut_index(windows[i].site_begin)];
            w.t1 = seg_tile_start[cut_index(windows[i].site_end)];
        } else {
            w.t0 = w.t1 = 0;
        }
    }
}

}  // namespace impop

using namespace impop;

struct impop_scan_plan {
    impop_ctx *ctx = nullptr;
    const impop_matrix *m = nullptr;
    uint64_t n_windows = 0, n_tiles = 0, bytes_streamed = 0;
    PopSizes ps{};
    bool subset_p = false;
    std::vector<uint32_t> masks;  // p | a | b, wps dwords each
    int d_pi_mode = 0, s_scope = 0;
    int finalize_tpw = 1;  // threads per window of scan_finalize_kernel: 1, 64 or 256 by the longest tile range
    ScanTile *d_tiles = nullptr;
    WinDesc *d_wins = nullptr;
    TilePartial *d_parts = nullptr;
    uint32_t *d_masks = nullptr;
    impop_window_stats *d_out = nullptr;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // pool, one pair per timed launch
    size_t events_used = 0;
};

// W of every window in ORIGINAL coordinates: its length, or the sum of its columns' weights
static int window_weights(const impop_matrix *m, const impop_window *windows, uint64_t n_windows, std::vector<WinDesc> &wd) {
    for (uint64_t i = 0; i < n_windows; ++i) {
        if (!m->wt_prefix.empty()) {
            wd[i].n_sites = m->wt_prefix[windows[i].site_end] - m->wt_prefix[windows[i].site_begin];
            // impop_window_stats.n_sites is 32 bits wide: refuse rather than truncate (unweighted windows are
            // checked against the same limit by their length)
            REQUIRE(wd[i].n_sites <= 0xFFFFFFFFull, "window %llu: the weights of its columns add up to %llu >= 2^32; split the window",
                    (unsigned long long)i, (unsigned long long)wd[i].n_sites);
        } else if (m->compact) {
            wd[i].n_sites = windows[i].site_end - windows[i].site_begin;
        }
    }
    return IMPOP_OK;
}

// Default tile: ~256 KB of matrix per workgroup, but never so large that a small job leaves CUs without
// work (>= 16 tiles per CU wanted), and never below the 32 blocks the kernel was tuned with.
static uint32_t default_tile_blocks(const impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows) {
    // ~256 KB per tile; wide sites (the any-n kernel, wps > 16) go down to 4 blocks = one per wave
    const uint32_t by_bytes = m->g.wps > 16 ? std::max<uint32_t>(4, 1024 / m->g.wps) : std::max<uint32_t>(32, 1024 / m->g.wps);
    uint64_t blocks = 0;
    for (uint64_t i = 0; i < n_windows; ++i) blocks += (windows[i].site_end - windows[i].site_begin + 63) / 64;
    if (m->compact && blocks > m->g.n_block) blocks = m->g.n_block;
    const uint64_t by_parallelism = blocks / (16ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256));
    return (uint32_t)std::max<uint64_t>(std::min<uint32_t>(32, by_bytes), std::min<uint64_t>(by_bytes, by_parallelism));
}

// subset masks of a plan; the overlap of A and B is removed from both (h-fst.py:181-185)
static void plan_set_masks(impop_scan_plan *p, const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b) {
    const uint32_t n = p->m->g.n_hap, wps = p->m->g.wps;
    std::vector<uint32_t> mp, ma, mb;
    mask_to_dwords(mask_p, n, wps, true, mp);
    mask_to_dwords(mask_a, n, wps, false, ma);
    mask_to_dwords(mask_b, n, wps, false, mb);
    for (uint32_t k = 0; k < wps; ++k) {
        const uint32_t ov = ma[k] & mb[k];
        ma[k] &= ~ov; mb[k] &= ~ov;
    }
    p->ps.n = n; p->ps.nP = popcount_vec(mp); p->ps.nA = popcount_vec(ma); p->ps.nB = popcount_vec(mb);
    p->subset_p = p->ps.nP != n;
    p->masks.clear();
    p->masks.reserve(3 * wps);
    p->masks.insert(p->masks.end(), mp.begin(), mp.end());
    p->masks.insert(p->masks.end(), ma.begin(), ma.end());
    p->masks.insert(p->masks.end(), mb.begin(), mb.end());
}

template <int WPS>
static void launch_scan_fixed(impop_scan_plan *p, hipStream_t st) {
    MaskArgs<WPS> mk;
    for (int k = 0; k < WPS; ++k) {
        mk.p[k] = p->masks[k];
        mk.a[k] = p->masks[WPS + k];
        mk.b[k] = p->masks[2 * WPS + k];
    }
    if (p->subset_p)
        hipLaunchKernelGGL((scan_tiles_kernel<WPS, true>), dim3((uint32_t)p->n_tiles), dim3(256), 0, st, p->m->d_sb,
                           p->d_tiles, mk, p->ps, p->d_parts);
    else
        hipLaunchKernelGGL((scan_tiles_kernel<WPS, false>), dim3((uint32_t)p->n_tiles), dim3(256), 0, st, p->m->d_sb,
                           p->d_tiles, mk, p->ps, p->d_parts);
}

IMPOP_API int impop_scan_plan_create(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows,
                                     uint64_t n_windows, const uint64_t *mask_p, const uint64_t *mask_a,
                                     const uint64_t *mask_b, const impop_scan_params *params, impop_scan_plan **out) {
    REQUIRE(ctx && m && out, "impop_scan_plan_create: NULL argument");
    *out = nullptr;
    REQUIRE(n_windows == 0 || windows, "impop_scan_plan_create: windows is NULL");
    REQUIRE(m->device == ctx->device, "impop_scan_plan_create: matrix lives on device %d, context on %d", m->device,
            ctx->device);
    REQUIRE(m->g.n_hap <= 65535, "impop_scan: n_hap %u > 65535 not supported by the 32-bit per-site products",
            m->g.n_hap);
    impop_scan_params prm;
    prm.struct_size = sizeof prm; prm.d_pi_mode = 0; prm.s_scope = 0; prm.tile_blocks = 0;
    if (params) {
        REQUIRE(params->struct_size == sizeof prm, "impop_scan_params.struct_size %u != %zu", params->struct_size, sizeof prm);
        prm = *params;
    }
    REQUIRE(prm.d_pi_mode >= 0 && prm.d_pi_mode <= 2, "impop_scan_params.d_pi_mode must be 0..2");
    REQUIRE(prm.s_scope == 0 || prm.s_scope == 1, "impop_scan_params.s_scope must be 0 or 1");
    // default tile: ~160 KB of matrix per workgroup.  With few haplotypes a 32-block tile is only a few KB and
    // the per-workgroup costs (launch, LDS reduction, partial store) bound the kernel instead of HBM:
    // n = 32 ran at 2.6 TB/s with 32-block tiles and 5.0 TB/s with whole-window tiles
    uint32_t tile_blocks = prm.tile_blocks ? prm.tile_blocks : default_tile_blocks(ctx, m, windows, n_windows);
    REQUIRE(tile_blocks <= 4096, "impop_scan_params.tile_blocks too large");
    for (uint64_t i = 0; i < n_windows; ++i) {
        REQUIRE(windows[i].site_begin <= windows[i].site_end && windows[i].site_end <= matrix_span(m),
                "window %llu: bad site range [%llu,%llu) for %llu sites", (unsigned long long)i,
                (unsigned long long)windows[i].site_begin, (unsigned long long)windows[i].site_end,
                (unsigned long long)matrix_span(m));
        REQUIRE(windows[i].site_end - windows[i].site_begin <= 0xFFFFFFFFull, "window %llu longer than 2^32 sites",
                (unsigned long long)i);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    impop_scan_plan *p = new impop_scan_plan();
    p->ctx = ctx; p->m = m; p->n_windows = n_windows;
    m->users++;
    p->d_pi_mode = prm.d_pi_mode; p->s_scope = prm.s_scope;
    const uint32_t wps = m->g.wps;
    plan_set_masks(p, mask_p, mask_a, mask_b);

    std::vector<ScanTile> tiles;
    std::vector<WinDesc> wd;
    {
        std::vector<impop_window> mapped;  // compacted matrix: original coordinates -> kept-site index ranges
        map_windows(m, windows, n_windows, mapped);
        build_tiles(mapped.data(), n_windows, tile_blocks, wps, tiles, wd, p->bytes_streamed);
    }
    p->n_tiles = tiles.size();
    auto fail = [&](int code) {
        impop_scan_plan_destroy(p);
        return code;
    };
    {
        const int wrc = window_weights(m, windows, n_windows, wd);
        if (wrc) return fail(wrc);
    }
    uint64_t longest_range = 0;
    for (const WinDesc &w : wd) longest_range = std::max(longest_range, w.t1 - w.t0);
    p->finalize_tpw = longest_range > 2048 ? 256 : longest_range > 48 ? 64 : 1;
    if (p->n_tiles >= 0x7FFFFFFFull) {
        set_error("impop_scan: %llu tiles exceed one launch; raise tile_blocks", (unsigned long long)p->n_tiles);
        return fail(IMPOP_E_INVALID);
    }
    hipError_t e;
#define PLAN_TRY(expr) \
    if ((e = (expr)) != hipSuccess) return fail(hip_fail(e, #expr, __FILE__, __LINE__))
    PLAN_TRY(hipMalloc((void **)&p->d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(ScanTile)));
    PLAN_TRY(hipMalloc((void **)&p->d_parts, std::max<size_t>(tiles.size(), 1) * sizeof(TilePartial)));
    PLAN_TRY(hipMalloc((void **)&p->d_wins, std::max<size_t>(n_windows, 1) * sizeof(WinDesc)));
    PLAN_TRY(hipMalloc((void **)&p->d_out, std::max<size_t>(n_windows, 1) * sizeof(impop_window_stats)));
    PLAN_TRY(hipMalloc((void **)&p->d_masks, (size_t)3 * wps * 4));
    if (!tiles.empty()) PLAN_TRY(hipMemcpyAsync(p->d_tiles, tiles.data(), tiles.size() * sizeof(ScanTile), hipMemcpyHostToDevice, ctx->stream));
    if (n_windows) PLAN_TRY(hipMemcpyAsync(p->d_wins, wd.data(), wd.size() * sizeof(WinDesc), hipMemcpyHostToDevice, ctx->stream));
    PLAN_TRY(hipMemcpyAsync(p->d_masks, p->masks.data(), (size_t)3 * wps * 4, hipMemcpyHostToDevice, ctx->stream));
    PLAN_TRY(hipStreamSynchronize(ctx->stream));  // host vectors die at return
#undef PLAN_TRY
    int rc = ensure_tajima_consts(ctx, p->ps.nP >= 2 ? (int64_t)p->ps.nP : 2);
    if (rc) return fail(rc);
    *out = p;
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_set_masks(impop_scan_plan *p, const uint64_t *mask_p, const uint64_t *mask_a,
                                        const uint64_t *mask_b) {
    REQUIRE(p, "impop_scan_plan_set_masks: plan is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    // launches already queued read the previous masks: kernel arguments were captured at their launch and
    // the device copy is only rewritten behind them in stream order
    plan_set_masks(p, mask_p, mask_a, mask_b);
    HIP_TRY(hipMemcpyAsync(p->d_masks, p->masks.data(), p->masks.size() * 4, hipMemcpyHostToDevice, p->ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_launch(impop_scan_plan *p, void *d_out) {
    REQUIRE(p, "impop_scan_plan_launch: plan is NULL");
    impop_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    // one process may drive several devices (impop_scan_sharded): kernels go to the device the plan's stream lives on
    HIP_TRY(hipSetDevice(ctx->device));
    // the cached Tajima constants belong to the context; another plan may have changed n since
    int rc = ensure_tajima_consts(ctx, p->ps.nP >= 2 ? (int64_t)p->ps.nP : 2);
    if (rc) return rc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (p->timing && p->n_tiles) {
        if (p->events_used == p->events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            p->events.push_back({a, b});
        }
        ev0 = p->events[p->events_used].first;
        ev1 = p->events[p->events_used].second;
        p->events_used++;
        HIP_TRY(hipEventRecord(ev0, st));
    }
    const bool weighted = !p->m->wt_prefix.empty();  // W of each window came from the host prefix sums at plan time
    const uint32_t wps = p->m->g.wps;
    if (p->n_tiles && (weighted || wps > 16)) {
        const size_t lds = (size_t)3 * ((wps + 3) & ~3u) * 4;
#define ANYN(SP, WT)                                                                                                     \
    hipLaunchKernelGGL((scan_tiles_anyn_kernel<SP, WT>), dim3((uint32_t)p->n_tiles), dim3(256), lds, st, p->m->d_sb,     \
                       p->d_tiles, p->d_masks, wps, p->m->g.G, p->m->g.r, p->ps, p->m->d_wt, p->d_parts)
        if (weighted) { if (p->subset_p) ANYN(true, true); else ANYN(false, true); }
        else          { if (p->subset_p) ANYN(true, false); else ANYN(false, false); }
#undef ANYN
        HIP_TRY(hipGetLastError());
        if (ev1) HIP_TRY(hipEventRecord(ev1, st));
    } else if (p->n_tiles) {
        switch (wps) {
#define CASE(W) case W: launch_scan_fixed<W>(p, st); break;
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
            CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
        }
        HIP_TRY(hipGetLastError());
        if (ev1) HIP_TRY(hipEventRecord(ev1, st));
    }
    if (p->n_windows) {
        impop_window_stats *dst = d_out ? (impop_window_stats *)d_out : p->d_out;
        if (p->finalize_tpw == 1)
            hipLaunchKernelGGL(scan_finalize_kernel<1>, dim3((uint32_t)((p->n_windows + 127) / 128)), dim3(128), 0, st, p->d_parts,
                               p->d_wins, p->n_windows, p->ps, ctx->d_taj, p->d_pi_mode, p->s_scope, dst);
        else if (p->finalize_tpw == 64)
            hipLaunchKernelGGL(scan_finalize_kernel<64>, dim3((uint32_t)((p->n_windows + 3) / 4)), dim3(256), 0, st, p->d_parts,
                               p->d_wins, p->n_windows, p->ps, ctx->d_taj, p->d_pi_mode, p->s_scope, dst);
        else
            hipLaunchKernelGGL(scan_finalize_kernel<256>, dim3((uint32_t)p->n_windows), dim3(256), 0, st, p->d_parts,
                               p->d_wins, p->n_windows, p->ps, ctx->d_taj, p->d_pi_mode, p->s_scope, dst);
        HIP_TRY(hipGetLastError());
    }
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_fetch(impop_scan_plan *p, impop_window_stats *out_host) {
    REQUIRE(p && (out_host || p->n_windows == 0), "impop_scan_plan_fetch: NULL argument");
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (p->n_windows)
        HIP_TRY(hipMemcpyAsync(out_host, p->d_out, p->n_windows * sizeof(impop_window_stats), hipMemcpyDeviceToHost,
                               p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_device_records(impop_scan_plan *p, void **d_records) {
    REQUIRE(p && d_records, "impop_scan_plan_device_records: NULL argument");
    *d_records = p->d_out;
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_info(const impop_scan_plan *p, uint64_t *n_tiles, uint64_t *bytes_streamed) {
    REQUIRE(p, "impop_scan_plan_info: plan is NULL");
    if (n_tiles) *n_tiles = p->n_tiles;
    if (bytes_streamed) *bytes_streamed = p->bytes_streamed;
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_timing(impop_scan_plan *p, int enable) {
    REQUIRE(p, "impop_scan_plan_timing: plan is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->timing = enable != 0;
    p->events_used = 0;
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_elapsed(impop_scan_plan *p, double *total_ms, uint64_t *launches) {
    REQUIRE(p, "impop_scan_plan_elapsed: plan is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    double t = 0.0;
    for (size_t i = 0; i < p->events_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->events[i].first, p->events[i].second));
        t += (double)ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = p->events_used;
    return IMPOP_OK;
}

IMPOP_API int impop_scan_plan_destroy(impop_scan_plan *p) {
    if (!p) return IMPOP_OK;
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    for (auto &e : p->events) {
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    if (p->d_tiles) hipFree(p->d_tiles);
    if (p->d_parts) hipFree(p->d_parts);
    if (p->d_wins) hipFree(p->d_wins);
    if (p->d_out) hipFree(p->d_out);
    if (p->d_masks) hipFree(p->d_masks);
    if (p->m) p->m->users--;
    delete p;
    return IMPOP_OK;
}

IMPOP_API int impop_scan(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                         const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b,
                         const impop_scan_params *params, impop_window_stats *out_host) {
    impop_scan_plan *p = nullptr;
    int rc = impop_scan_plan_create(ctx, m, windows, n_windows, mask_p, mask_a, mask_b, params, &p);
    if (rc) return rc;
    rc = impop_scan_plan_launch(p, nullptr);
    if (!rc) rc = impop_scan_plan_fetch(p, out_host);
    impop_scan_plan_destroy(p);
    return rc;
}

IMPOP_API int impop_site_counts(impop_ctx *ctx, const impop_matrix *m, const uint64_t *mask, uint64_t site_begin,
                                uint64_t site_end, uint32_t *out_host) {
    REQUIRE(ctx && m, "impop_site_counts: NULL argument");
    NOT_COMPACT(m, "impop_site_counts");
    REQUIRE(site_begin <= site_end && site_end <= m->g.n_site, "impop_site_counts: bad site range");
    const uint64_t W = site_end - site_begin;
    if (!W) return IMPOP_OK;
    REQUIRE(out_host, "impop_site_counts: out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint32_t> mk;
    mask_to_dwords(mask, m->g.n_hap, m->g.wps, true, mk);
    void *d = nullptr;
    const size_t mask_bytes = ((size_t)m->g.wps * 4 + 255) / 256 * 256;
    int rc = ctx_scratch(ctx, mask_bytes + W * 4, &d);
    if (rc) return rc;
    uint32_t *d_mask = (uint32_t *)d, *d_cnt = (uint32_t *)((char *)d + mask_bytes);
    HIP_TRY(hipMemcpyAsync(d_mask, mk.data(), (size_t)m->g.wps * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t nb = (site_end + 63) / 64 - site_begin / 64;
    REQUIRE((nb + 3) / 4 < 0x7FFFFFFFull, "impop_site_counts: range too long");
    hipLaunchKernelGGL(site_counts_kernel, dim3((uint32_t)((nb + 3) / 4)), dim3(256), 0, ctx->stream, m->d_sb, d_mask,
                       m->g.wps, m->g.G, m->g.r, site_begin, site_end, d_cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, d_cnt, W * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

template <int K>
static int launch_multi(hipStream_t st, const impop_matrix *m, uint64_t n_tiles, const ScanTile *d_tiles, const uint32_t *d_masks,
                        const uint32_t *d_n, uint64_t *d_parts, bool small) {
    const size_t lds = (size_t)K * ((m->g.wps + 3) & ~3u) * 4;
    if (small) {
        hipLaunchKernelGGL((scan_multi_kernel<K, true>), dim3((uint32_t)n_tiles), dim3(256), lds, st, m->d_sb, d_tiles, d_masks, d_n,
                           m->g.wps, m->g.G, m->g.r, m->d_wt, d_parts);
    } else {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)scan_multi_kernel<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((scan_multi_kernel<K, false>), dim3((uint32_t)n_tiles), dim3(256), lds, st, m->d_sb, d_tiles, d_masks, d_n,
                           m->g.wps, m->g.G, m->g.r, m->d_wt, d_parts);
    }
    return IMPOP_OK;
}

IMPOP_API int impop_scan_multi(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                               const uint64_t *masks, uint32_t n_pop, impop_pair_stats *out_host) {
    REQUIRE(ctx && m, "impop_scan_multi: NULL argument");
    REQUIRE(n_pop >= 2 && n_pop <= 8, "impop_scan_multi: n_pop must be 2..8");
    REQUIRE(masks, "impop_scan_multi: masks is NULL");
    REQUIRE(m->g.n_hap <= 65535, "impop_scan_multi: n_hap > 65535 not supported");
    if (!n_windows) return IMPOP_OK;
    REQUIRE(windows && out_host, "impop_scan_multi: NULL windows/out");
    for (uint64_t i = 0; i < n_windows; ++i)
        REQUIRE(windows[i].site_begin <= windows[i].site_end && windows[i].site_end <= matrix_span(m) &&
                    windows[i].site_end - windows[i].site_begin <= 0xFFFFFFFFull,
                "impop_scan_multi: window %llu: bad site range", (unsigned long long)i);
    const uint32_t n = m->g.n_hap, wps = m->g.wps, K = n_pop, NP = K * (K - 1) / 2;
    const uint32_t mwords = (n + 63) / 64;
    std::vector<uint32_t> mk((size_t)K * wps), nk(K);
    std::vector<uint32_t> seen(wps, 0u), one;
    for (uint32_t k = 0; k < K; ++k) {
        mask_to_dwords(masks + (size_t)k * mwords, n, wps, false, one);
        for (uint32_t j = 0; j < wps; ++j) {
            // h-fst.py:181-185 removes shared members per pair; with K populations that would make
            // n_k pair-dependent, so the one-pass form requires disjoint populations
            REQUIRE((seen[j] & one[j]) == 0, "impop_scan_multi: populations must be disjoint (population %u overlaps an earlier one)", k);
            seen[j] |= one[j];
            mk[(size_t)k * wps + j] = one[j];
        }
        nk[k] = popcount_vec(one);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<ScanTile> tiles;
    std::vector<WinDesc> wd;
    uint64_t bytes = 0;
    const uint32_t tile_blocks_used = default_tile_blocks(ctx, m, windows, n_windows);
    {
        std::vector<impop_window> mapped;
        map_windows(m, windows, n_windows, mapped);
        build_tiles(mapped.data(), n_windows, tile_blocks_used, wps, tiles, wd, bytes);
        const int wrc = window_weights(m, windows, n_windows, wd);
        if (wrc) return wrc;
    }
    REQUIRE(tiles.size() < 0x7FFFFFFFull, "impop_scan_multi: too many tiles");
    const size_t nt = tiles.size();
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_tiles = 0, o_wins = o_tiles + up(std::max<size_t>(nt, 1) * sizeof(ScanTile)),
                 o_masks = o_wins + up(n_windows * sizeof(WinDesc)), o_n = o_masks + up(mk.size() * 4),
                 o_parts = o_n + up(K * 4), o_out = o_parts + up(std::max<size_t>(nt, 1) * (K + NP) * 8),
                 total = o_out + up(n_windows * NP * sizeof(impop_pair_stats));
    void *d = nullptr;
    int rc = ctx_scratch(ctx, total, &d);
    if (rc) return rc;
    char *base = (char *)d;
    if (nt) HIP_TRY(hipMemcpyAsync(base + o_tiles, tiles.data(), nt * sizeof(ScanTile), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + o_wins, wd.data(), n_windows * sizeof(WinDesc), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + o_masks, mk.data(), mk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + o_n, nk.data(), K * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nt) {
        const ScanTile *dt = (const ScanTile *)(base + o_tiles);
        const uint32_t *dm = (const uint32_t *)(base + o_masks), *dn = (const uint32_t *)(base + o_n);
        uint64_t *dp = (uint64_t *)(base + o_parts);
        // 32-bit per-lane partial sums: unweighted, <= 512 haplotypes, <= 1024 sites per lane and tile
        const bool small = m->wt_prefix.empty() && n <= 512 && tile_blocks_used <= 4096;
        switch (K) {
            case 2: rc = launch_multi<2>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            case 3: rc = launch_multi<3>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            case 4: rc = launch_multi<4>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            case 5: rc = launch_multi<5>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            case 6: rc = launch_multi<6>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            case 7: rc = launch_multi<7>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
            default: rc = launch_multi<8>(ctx->stream, m, nt, dt, dm, dn, dp, small); break;
        }
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
    }
    const uint64_t items = n_windows * NP;
    hipLaunchKernelGGL(scan_multi_finalize_kernel, dim3((uint32_t)((items + 127) / 128)), dim3(128), 0, ctx->stream,
                       (const uint64_t *)(base + o_parts), (const WinDesc *)(base + o_wins), n_windows, K,
                       (const uint32_t *)(base + o_n), (impop_pair_stats *)(base + o_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, base + o_out, items * sizeof(impop_pair_stats), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_afs(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                        const uint64_t *mask, uint32_t *out_host) {
    REQUIRE(ctx && m, "impop_afs: NULL argument");
    NOT_COMPACT(m, "impop_afs");
    if (!n_windows) return IMPOP_OK;
    REQUIRE(windows && out_host, "impop_afs: NULL windows/out");
    uint64_t longest = 0;
    for (uint64_t i = 0; i < n_windows; ++i) {
        REQUIRE(windows[i].site_begin <= windows[i].site_end && windows[i].site_end <= m->g.n_site,
                "impop_afs: window %llu: bad site range", (unsigned long long)i);
        longest = std::max(longest, windows[i].site_end - windows[i].site_begin);
    }
    std::vector<uint32_t> mk;
    mask_to_dwords(mask, m->g.n_hap, m->g.wps, true, mk);
    const uint32_t bins = popcount_vec(mk) + 1;
    REQUIRE((size_t)bins * 4 <= 64 * 1024, "impop_afs: more than 16383 haplotypes in the mask");
    HIP_TRY(hipSetDevice(ctx->device));
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_mask = 0, o_wins = up((size_t)m->g.wps * 4), o_out = o_wins + up(n_windows * sizeof(impop_window)),
                 total = o_out + n_windows * bins * 4;
    void *d = nullptr;
    int rc = ctx_scratch(ctx, total, &d);
    if (rc) return rc;
    char *base = (char *)d;
    HIP_TRY(hipMemcpyAsync(base + o_mask, mk.data(), (size_t)m->g.wps * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + o_wins, windows, n_windows * sizeof(impop_window), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(base + o_out, 0, n_windows * bins * 4, ctx->stream));
    // a workgroup per window when there are thousands of them (one flush of the histogram per window, no atomics), more —
    // never shorter than 4096 sites — when few windows would leave CUs idle (about 32 workgroups per CU wanted)
    const uint64_t want = 32ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    uint64_t chunks = std::min<uint64_t>((longest + 4095) / 4096, std::max<uint64_t>(1, (want + n_windows - 1) / std::max<uint64_t>(n_windows, 1)));
    const uint64_t chunk_sites = chunks ? ((longest + chunks - 1) / chunks + 63) / 64 * 64 : 0;
    if (chunks) chunks = (longest + chunk_sites - 1) / chunk_sites;
    if (chunks) {
        REQUIRE(chunks < 0x7FFFFFFFull, "impop_afs: window too long");
        // windows ride on gridDim.y (<= 65535): any number of windows goes out in batches of that many
        for (uint64_t w0 = 0; w0 < n_windows; w0 += 65535) {
            const uint32_t nw = (uint32_t)std::min<uint64_t>(65535, n_windows - w0);
            hipLaunchKernelGGL(afs_kernel, dim3((uint32_t)chunks, nw), dim3(256), (size_t)bins * 4, ctx->stream, m->d_sb,
                               (const uint32_t *)(base + o_mask), m->g.wps, m->g.G, m->g.r,
                               (const impop_window *)(base + o_wins) + w0, bins, chunk_sites, (uint32_t *)(base + o_out) + w0 * bins);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipMemcpyAsync(out_host, base + o_out, n_windows * bins * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}
