// stats_kernels.h — the reference's identity-matrix statistics (pica2.py, h-fst.py, af.py,
// tj_d.py) as one-workgroup-per-problem HIP kernels.  A "problem" is one identity matrix:
// either dense doubles (the .sim drop-in path) or the integer Gram counts of a window
// (pairwise.hip), from which identities are formed on the fly (never stored in HBM).
#pragma once
#include "device_utils.h"
#include "internal.h"

namespace impop {

struct SimBatch {
    const double *dense;   // problem p at dense + p*stride (n x ld doubles) or nullptr
    const int32_t *gram;   // problem p at gram + p*stride (ld x ld int32) or nullptr
    uint64_t stride;       // elements between consecutive problems
    uint32_t ld;
    uint32_t n;            // elements of the full matrix
    const uint64_t *W;     // per-problem site count (gram mode)
    int kind;
    int round_digits;
    // gram mode, optional: problem p is the SUM of the consecutive Gram matrices seg_first[p] .. +seg_count[p]
    // (elementary segments shared by overlapping windows: I_ij is additive over disjoint site ranges);
    // nullptr = exactly one matrix per problem, at index p
    const uint32_t *seg_first;
    const uint32_t *seg_count;
    // gram mode, optional: per-problem constant added to EVERY I_ij (diagonal included): the sites all haplotypes
    // carry, which a compacted matrix dropped (pairwise.hip)
    const uint32_t *add;
    uint32_t *err;  // optional device error word (internal.h DEV_ERR_*): set by the launch_* functions
    uint32_t g16;   // gram mode: the counts are uint16 (pairwise.hip writes them so when every window's W < 65536: half the bytes)
    uint64_t max_W; // gram mode, optional: no problem's W exceeds this (0 = not stated); lets a launch pick 32-bit Hamming arithmetic
};

struct SimView {
    const double *dense;
    const int32_t *gram;
    uint32_t ld;
    uint64_t W;
    int kind;
    int round_digits;
    // optional memo in LDS for the `match` identity of a Gram problem: identity is a function of the
    // Hamming distance H alone, so (W - H)/W and its CPython rounding are tabulated once per window
    // for H < tbl_n (same arithmetic, hence bit-identical values) instead of once per pair
    const double *tbl;
    uint32_t tbl_n;
    const int32_t *diag;  // optional LDS copy of the Gram diagonal a_i (else read from `gram`)
    uint32_t nseg;        // Gram matrices to add up (gram mode)
    uint64_t seg_stride;  // elements between them
    int64_t add;          // constant added to every Gram entry
    uint32_t *err;        // device error word or nullptr
    uint32_t g16;         // the Gram counts behind `gram` are uint16 (read them with gram_ld1 / gram_ld4 only)
};

constexpr uint32_t SIM_TBL_N = 1024;  // 8 KB: LDS footprint decides the occupancy of the epilogue kernels

__device__ __forceinline__ double match_identity(uint64_t W, int64_t H, int round_digits) {
    double v = W ? (double)((int64_t)W - H) / (double)W : 1.0;
    if (round_digits >= 0) v = py_round(v, round_digits);
    return v;
}

// cooperative fill by the whole workgroup (call before any sim_get; caller synchronises)
__device__ __forceinline__ void sim_table_fill(SimView &S, double *lds_tbl, uint32_t n_threads) {
    S.tbl = nullptr;
    S.tbl_n = 0;
    if (S.gram && S.kind == IMPOP_IDENTITY_MATCH) {
        for (uint32_t h = threadIdx.x; h < SIM_TBL_N; h += n_threads) lds_tbl[h] = match_identity(S.W, (int64_t)h, S.round_digits);
        S.tbl = lds_tbl;
        S.tbl_n = SIM_TBL_N;
    }
}

__device__ __forceinline__ SimView sim_view(const SimBatch &b, uint64_t p) {
    SimView v;
    v.dense = b.dense ? b.dense + p * b.stride : nullptr;
    v.nseg = 1;
    v.seg_stride = b.stride;
    const uint64_t first = (b.gram && b.seg_first) ? (uint64_t)b.seg_first[p] : p;  // index of the problem's (first) Gram matrix
    if (b.gram && b.seg_first) v.nseg = b.seg_count[p];
    v.g16 = b.gram ? b.g16 : 0;
    v.gram = !b.gram ? nullptr
             : b.g16 ? reinterpret_cast<const int32_t *>(reinterpret_cast<const uint16_t *>(b.gram) + first * b.stride)
                     : b.gram + first * b.stride;
    v.ld = b.ld;
    v.W = b.W ? b.W[p] : 0;
    v.kind = b.kind;
    v.round_digits = b.round_digits;
    v.tbl = nullptr;
    v.tbl_n = 0;
    v.diag = nullptr;
    v.add = (b.gram && b.add) ? (int64_t)b.add[p] : 0;
    v.err = b.err;
    return v;
}

// element e of the problem's first Gram matrix (segment k: e + k * seg_stride), 32- or 16-bit storage
typedef int gram_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short gram_u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int32_t gram_ld1(const SimView &S, uint64_t e) {
    return S.g16 ? (int32_t)reinterpret_cast<const uint16_t *>(S.gram)[e] : S.gram[e];
}
__device__ __forceinline__ gram_i32x4 gram_ld4(const SimView &S, uint64_t e) {  // four consecutive elements, e a multiple of 4
    if (S.g16) {
        const gram_u16x4 v = *reinterpret_cast<const gram_u16x4 *>(reinterpret_cast<const uint16_t *>(S.gram) + e);
        return gram_i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
    return *reinterpret_cast<const gram_i32x4 *>(S.gram + e);
}
__device__ __forceinline__ bool gram_quads_aligned(const SimView &S) {  // may rows be read four elements at a time?
    return (S.ld & 3u) == 0 && (S.seg_stride & 3ull) == 0 && ((uintptr_t)S.gram & (S.g16 ? 7 : 15)) == 0;
}
// Gram entry (i, j) of a problem = sum over its segments
__device__ __forceinline__ int64_t gram_at(const SimView &S, uint32_t i, uint32_t j) {
    const uint64_t e = (uint64_t)i * S.ld + j;
    int64_t v = S.add;
    for (uint32_t k = 0; k < S.nseg; ++k) v += gram_ld1(S, e + k * S.seg_stride);
    return v;
}

// identity from the Gram counts of a pair: I = shared sites, ai / aj = sites carried by i / by j
__device__ __forceinline__ double sim_from_gram(const SimView &S, int64_t I, int64_t ai, int64_t aj) {
    double v;
    if (S.kind == IMPOP_IDENTITY_MATCH) {
        const int64_t H = ai + aj - 2 * I;
        if ((uint64_t)H < S.tbl_n) return S.tbl[H];  // memoised (already rounded)
        v = S.W ? (double)((int64_t)S.W - H) / (double)S.W : 1.0;
    } else {
        const int64_t d = ai + aj;
        v = d ? (double)(2 * I) / (double)d : 1.0;
    }
    if (S.round_digits >= 0) v = py_round(v, S.round_digits);
    return v;
}

// identity of the unordered pair {i, j}; NaN = pair absent (pica2.py:85-87 keying)
__device__ __forceinline__ double sim_get(const SimView &S, uint32_t i, uint32_t j) {
    if (i > j) { const uint32_t t = i; i = j; j = t; }
    if (!S.dense)
        return sim_from_gram(S, gram_at(S, i, j), S.diag ? S.diag[i] : gram_at(S, i, i), S.diag ? S.diag[j] : gram_at(S, j, j));
    double v = S.dense[(uint64_t)i * S.ld + j];
    if (S.round_digits >= 0 && v == v) v = py_round(v, S.round_digits);
    return v;
}

struct Pica2Out {
    double pi, pi_site;
    uint32_t n_groups, pad;
    double sum_2pairs;   // sum(2 * pair for pair in group_pairs), pica2.py:154/159
    uint64_t n_pairs;    // len(group_pairs), pica2.py:158
};

struct HfstOut {
    double v[6];       // fst, pi_a, pi_b, pi_xy, dxy, da
    uint64_t cnt[6];   // pairs_a, miss_a, pairs_b, miss_b, pairs_between, miss_between
};

// h-fst.py:168-240: the six outputs from the three sums and the pair / missing-pair counts
__device__ __forceinline__ void hfst_outputs(double accA, double accB, double accX, uint64_t cA, uint64_t mA, uint64_t cB, uint64_t mB,
                                    uint64_t cX, uint64_t mX, uint64_t L, HfstOut *__restrict__ dst) {
    const double pi_a = cA ? accA / (double)cA : 0.0;  // h-fst.py:168-171
    const double pi_b = cB ? accB / (double)cB : 0.0;
    const double dxy = cX ? accX / (double)cX : 0.0;
    const double pi_xy = 0.5 * (pi_a + pi_b);                   // :203
    const double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;   // :214-221
    HfstOut o;
    if (L > 0) {  // :225-240
        const double dl = (double)L;
        o.v[0] = fst; o.v[1] = pi_a / dl; o.v[2] = pi_b / dl; o.v[3] = pi_xy / dl; o.v[4] = dxy / dl;
        o.v[5] = (dxy - pi_xy) / dl;
    } else {
        o.v[0] = fst; o.v[1] = pi_a; o.v[2] = pi_b; o.v[3] = pi_xy; o.v[4] = dxy; o.v[5] = dxy - pi_xy;
    }
    o.cnt[0] = cA; o.cnt[1] = mA; o.cnt[2] = cB; o.cnt[3] = mB; o.cnt[4] = cX; o.cnt[5] = mX;
    *dst = o;
}

// the window-statistics shape (Gram problems of <= 512 sequences, `match`, uint16 counts, one matrix per problem): stats_small.hip
bool pica2_small_applies(const SimBatch &b, uint32_t n_el, const uint32_t *d_order, const uint32_t *d_group_of);
bool hfst_small_applies(const SimBatch &b);
int launch_pica2_small(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_idx, uint32_t n_el, double threshold,
                       const uint64_t *d_seq_len, Pica2Out *d_out);
int launch_hfst_small(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint8_t *d_in_a, const uint8_t *d_in_b,
                      const uint64_t *d_seq_len, HfstOut *d_out);

// d_order (nullable): seed order of the greedy grouping as positions into the element list (stats.hip greedy_groups)
int launch_pica2(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_idx, uint32_t n_el,
                 const uint32_t *d_order, double threshold, const uint64_t *d_seq_len, Pica2Out *d_out,
                 uint32_t *d_group_of);
int launch_hfst(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint8_t *d_in_a, const uint8_t *d_in_b,
                const uint64_t *d_seq_len, HfstOut *d_out);
// hud.py grouped Fst: members of A / B as index lists (overlap already removed)
int launch_hud_grouped(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_ia, uint32_t ma,
                       const uint32_t *d_ib, uint32_t mb, const uint32_t *d_order_a, const uint32_t *d_order_b,
                       double threshold, const uint64_t *d_seq_len, HfstOut *d_out);
int launch_af(impop_ctx *ctx, const SimBatch &b, double threshold, uint32_t *d_adj, uint32_t *d_cluster_of,
              uint32_t *d_sizes, uint32_t *d_nclusters);

}  // namespace impop
