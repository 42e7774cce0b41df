// stats_small.hip — pica2 (pica2.py:60-169) and h-fst (h-fst.py:130-249) on the WINDOW-STATISTICS shape of the all-pairs
// path: Gram problems of up to 512 sequences, `match` identity, counts of 16 or 32 bits (W < 2^30), one Gram matrix per problem or a sum of segments — what
// impop_pairwise_scan hands over for disjoint windows of a few hundred haplotypes (BASELINE configs 1-4).  Same decisions and
// the same identities as the general kernels of stats.hip (which keep every other shape: dense .sim problems, `dice`, handed-in
// seed orders, windows as heavy as 2^30, thousands of sequences); what differs is the data layout.
//
// The general kernels walk a row four positions per lane (16-byte loads) and keep sets of positions in an interleaved bit
// layout; their per-pair work is tens of instructions and every load sits behind its own wait.  Counters over 4096 windows x
// 465 haplotypes (profiles/r03_epilogue_pmc.txt): pica2_kernel 212 k vector instructions per window at 465 groups, the vector
// unit 43 % busy, three quarters of the waves' cycles parked on memory; hfst_kernel 28 k instructions for 28.7 k pairs.
//
// Here lane L of a wave owns the FIXED positions L, L + 64, ... (8 per lane cover 512): what depends on the column alone —
// its element, its diagonal count a_j, its frequency, its class — is loaded into registers once per wave, a row is eight
// 2-byte loads (128 contiguous bytes per wave and load; all of a row's loads go out before the first is consumed), a pair is
// the Hamming distance H = a_i + a_j - 2 I_ij (the constant a compacted matrix adds to every count cancels), one compare or
// one table look-up, and `ballot` of a compare IS a 64-bit word of a position set in natural order.  The greedy grouping then
// runs on plain 512-bit sets: the free set in scalar registers, a block of up to 64 candidate rows in the lanes of one wave.
#include <stdlib.h>

#include "stats_kernels.h"

namespace impop {

namespace {

constexpr uint32_t SM_T = 256;        // threads per problem
constexpr uint32_t SM_N = 512;        // positions a problem may have
constexpr uint32_t SM_W = SM_N / 64;  // 64-bit words of a position set

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l /*uniform*/) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double block_sum(double v, double *sh /*SM_T / 64*/) {  // wave butterfly, then waves in order
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (uint32_t w = 0; w < SM_T / 64; ++w) t += sh[w];
    return t;
}
// skipped_load: a row's loads sit behind uniform branches (words left of the row, or past the list, are not fetched).  The value
// a skipped word keeps must NOT be a constant: with `0` the compiler folds the first use (2 * I) into the branch — next to the
// load, with a wait for it — and the row's loads go out one at a time instead of together.  A per-lane value nothing reads stops that.
// memoised `match` identity of a Hamming distance (stats_kernels.h sim_from_gram: same table, same arithmetic beyond it)
__device__ __forceinline__ double ident_of(const SimView &S, const double *tbl, int32_t H) {
    return (uint32_t)H < SIM_TBL_N ? tbl[H] : match_identity(S.W, (int64_t)H, S.round_digits);
}

// The counts of a batch of U rows: I[u][k] = the problem's I(row u, this lane's position in word k), for the words k0[u] .. nw a row
// needs (the others keep a value nobody reads).  off(u, k) = element offset of that entry in a Gram matrix.
//  one matrix per problem (SEG false): all loads of the batch go out before the first is consumed;
//  a problem that is the SUM of nseg consecutive matrices (sliding windows sharing elementary segments, stats_kernels.h SimBatch):
//  row by row, two segments' loads in flight, added up once both have been issued (nseg == 0: the counts are 0).
template <uint32_t NWK, int U, bool SEG, typename CT, class OFF>
__device__ __forceinline__ void row_counts(const CT *__restrict__ g, uint32_t nseg, uint32_t sstride, const bool (&lv)[U],
                                           const uint32_t (&k0)[U], uint32_t nw, int32_t junk, OFF off, int32_t (&I)[U][NWK]) {
    if (!SEG) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (uint32_t k = 0; k < NWK; ++k) {
                I[u][k] = junk;  // see skipped_load
                if (lv[u] && k >= k0[u] && k < nw) I[u][k] = (int32_t)g[off(u, k)];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (uint32_t k = 0; k < NWK; ++k) I[u][k] = 0;
            if (!lv[u]) continue;
            for (uint32_t s0 = 0; s0 < nseg; s0 += 2) {
                const bool two = s0 + 1 < nseg;
                // (a segment's base in 64 bits, uniform: ld^2 elements per matrix times hundreds of segments passes 2^32)
                const CT *__restrict__ g0 = g + (uint64_t)s0 * sstride, *__restrict__ g1 = g + (uint64_t)(two ? s0 + 1 : s0) * sstride;
                int32_t T0[NWK], T1[NWK];
#pragma unroll
                for (uint32_t k = 0; k < NWK; ++k) {
                    T0[k] = junk; T1[k] = junk;
                    if (k >= k0[u] && k < nw) {
                        const uint32_t o = off(u, k);
                        T0[k] = (int32_t)g0[o];
                        if (two) T1[k] = (int32_t)g1[o];
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < NWK; ++k)
                    if (k >= k0[u] && k < nw) I[u][k] += T0[k] + (two ? T1[k] : 0);
            }
        }
    }
}
// one entry of the problem (all segments)
template <bool SEG, typename CT>
__device__ __forceinline__ int32_t one_count(const CT *__restrict__ g, uint32_t nseg, uint32_t sstride, uint32_t off) {
    if (!SEG) return (int32_t)g[off];
    int32_t v = 0;
    for (uint32_t s = 0; s < nseg; ++s) v += (int32_t)(g + (uint64_t)s * sstride)[off];
    return v;
}

// "identity > threshold" is "H <= H*" (stats.hip match_cutoff: the largest H whose identity — the very function the pairs
// would be tested with — exceeds the threshold; -1: none).  One wave searches 64 distances at a time: three rounds for a
// 50 kb window instead of seventeen bisection steps every thread repeats.
__device__ __forceinline__ int32_t match_cutoff_wave(const SimView &S, double thr) {
    const uint32_t lane = threadIdx.x & 63;
    auto above = [&](int64_t H) { const double v = match_identity(S.W, H, S.round_digits); return v == v && v > thr; };
    const int64_t W = (int64_t)S.W;
    const bool a0 = above(0), aW = above(W);
    if (!a0) return -1;
    if (aW) return (int32_t)W;
    int64_t lo = 0, hi = W;  // above(lo), !above(hi)
    while (hi - lo > 1) {
        const int64_t step = (hi - lo + 63) / 64;
        const int64_t x = lo + (int64_t)(lane + 1) * step;
        const bool ok = x < hi && above(x);
        const int64_t t = (int64_t)__popcll(__ballot(ok));  // identity does not increase with H: the true lanes are a prefix
        const int64_t nhi = lo + (t + 1) * step;
        lo = lo + t * step;
        if (nhi < hi) hi = nhi;
    }
    return (int32_t)lo;
}

// ---------------------------------------------------------------------------------------
// h-fst.py:130-249 calculate_diversity / calculate_fst.  The members of the two classes (A only, B only: h-fst.py:181-185
// drops the overlap) are listed class-major — A ascending, then B ascending — and every unordered pair of members is
// visited once as (list index q, list index p > q): q < na, p < na is a pair inside A, q < na <= p a pair between, na <= q a
// pair inside B.  Lane L owns list indices L, L + 64, ...; a wave takes rows q and the words of the list right of them.
// NWK words of the list (4: up to 256 members, 8: up to 512), U rows in flight — U * NWK = 16 loads either way.
template <uint32_t NWK, int U, bool SEG, typename CT>
__device__ __forceinline__ void hfst_small_rows(const SimView &S, const double *tbl, const CT *__restrict__ g, uint32_t ld, uint32_t wave,
                                                uint32_t lane, uint32_t na, uint32_t nmem, uint32_t nw, const uint16_t *mpos,
                                                const int32_t *dg_l, double &accA, double &accB, double &accX) {
    const int32_t junk = (int32_t)lane;  // see skipped_load
    uint32_t pos[NWK];
    int32_t dgk[NWK];
#pragma unroll
    for (uint32_t k = 0; k < NWK; ++k) {
        const uint32_t p = 64 * k + lane;
        pos[k] = p < nmem ? (uint32_t)mpos[p] : 0u;
        dgk[k] = p < nmem ? dg_l[p] : 0;
    }
    double accRA[NWK];  // rows of class A: sums by word of the list (the word's class decides: inside A / between)
#pragma unroll
    for (uint32_t k = 0; k < NWK; ++k) accRA[k] = 0.0;
    // this wave's rows q = wave + 4 r, r = 0, 1, ...: what a row needs (position, diagonal count) is staged in the lanes once —
    // lane j of set t holds row r = j + 64 t — and comes back by v_readlane, so that nothing but the Gram loads themselves
    // stands between two batches of rows
    constexpr uint32_t NT = NWK / 4;  // staging sets: 64 rows per wave and set
    uint32_t prow[NT];
    int32_t arow[NT];
#pragma unroll
    for (uint32_t t = 0; t < NT; ++t) {
        const uint32_t q = wave + (SM_T / 64) * (lane + 64 * t);
        prow[t] = q < nmem ? (uint32_t)mpos[q] : 0u;
        arow[t] = q < nmem ? dg_l[q] : 0;
    }
    const uint32_t nrows = nmem > wave ? (nmem - wave + SM_T / 64 - 1) / (SM_T / 64) : 0;
    for (uint32_t r0 = 0; r0 < nrows; r0 += U) {
        uint32_t q[U], k0[U], pr[U];
        int32_t ar[U];
        bool lv[U];
        int32_t I[U][NWK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            lv[u] = r0 + u < nrows;
            const uint32_t r = lv[u] ? r0 + u : r0;
            q[u] = wave + (SM_T / 64) * r;
            k0[u] = q[u] >> 6;
            if (NT == 1 || r < 64) {
                pr[u] = (uint32_t)__builtin_amdgcn_readlane((int)prow[0], (int)(r & 63));
                ar[u] = __builtin_amdgcn_readlane(arow[0], (int)(r & 63));
            } else {
                pr[u] = (uint32_t)__builtin_amdgcn_readlane((int)prow[NT - 1], (int)(r & 63));
                ar[u] = __builtin_amdgcn_readlane(arow[NT - 1], (int)(r & 63));
            }
        }
        // the pair's entry in the upper triangle (the part the Gram kernel writes)
        row_counts<NWK, U, SEG, CT>(g, S.nseg, (uint32_t)S.seg_stride, lv, k0, nw, junk,
                                [&](int u, uint32_t k) { return (pr[u] < pos[k] ? pr[u] : pos[k]) * ld + (pr[u] < pos[k] ? pos[k] : pr[u]); }, I);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (uint32_t k = 0; k < NWK; ++k) {
                if (lv[u] && k >= k0[u] && k < nw) {
                    const uint32_t p = 64 * k + lane;
                    // a lane without a pair here (left of q, or past the list) looks up distance 0: identity 1, the term 0
                    const int32_t H = (p > q[u] && p < nmem) ? ar[u] + dgk[k] - 2 * I[u][k] : 0;
                    const double d = 1 - ident_of(S, tbl, H);
                    if (q[u] < na) accRA[k] += d; else accB += d;
                }
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < NWK; ++k) {
        if (64 * k + lane < na) accA += accRA[k]; else accX += accRA[k];
    }
}

template <bool SEG, typename CT>
__global__ __launch_bounds__(SM_T, SEG ? 4 : 5) void hfst_small_kernel(SimBatch batch, const uint8_t *__restrict__ in_a,
                                                             const uint8_t *__restrict__ in_b, const uint64_t *__restrict__ seq_len,
                                                             HfstOut *__restrict__ out) {
    __shared__ double tbl[SIM_TBL_N];
    __shared__ int32_t dg_l[SM_N];    // Gram diagonal of list member p
    __shared__ uint16_t mpos[SM_N];   // its position (= sequence index)
    __shared__ uint8_t cls_l[SM_N];   // class of sequence i: 1 = A only, 2 = B only, 0 = neither or both
    __shared__ uint32_t sh_na, sh_nb;
    __shared__ double shd[SM_T / 64];
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);
    S.dense = nullptr; S.g16 = sizeof(CT) == 2 ? 1u : 0u;  // (sim_view took the problem's base from batch.g16: the launch matches CT to it)
    if (!SEG) S.nseg = 1;
    const uint32_t n = batch.n, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), ld = S.ld;
    const CT *__restrict__ g = reinterpret_cast<const CT *>(S.gram);
    for (uint32_t i = tid; i < SM_N; i += SM_T) {  // (one round of loads for the workgroup; read one by one by the listing wave
        uint32_t cc = 0;                           //  they were sixteen dependent round trips at the head of every problem)
        if (i < n) {
            const bool a = in_a[i], b = in_b[i];
            cc = (a && !b) ? 1u : (b && !a) ? 2u : 0u;
        }
        cls_l[i] = (uint8_t)cc;
    }
    for (uint32_t h = tid; h < SIM_TBL_N; h += SM_T) tbl[h] = match_identity(S.W, (int64_t)h, S.round_digits);
    __syncthreads();
    if (wave == 0) {  // the member list, in order: ballots + prefix popcounts
        uint32_t na = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) na += (uint32_t)__popcll(__ballot(cls_l[i0 + lane] == 1));
        uint32_t pa = 0, pb = na;
        const uint64_t below = (1ull << lane) - 1ull;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t cc = cls_l[i0 + lane];
            const uint64_t balA = __ballot(cc == 1), balB = __ballot(cc == 2);
            if (cc == 1) mpos[pa + (uint32_t)__popcll(balA & below)] = (uint16_t)(i0 + lane);
            if (cc == 2) mpos[pb + (uint32_t)__popcll(balB & below)] = (uint16_t)(i0 + lane);
            pa += (uint32_t)__popcll(balA);
            pb += (uint32_t)__popcll(balB);
        }
        if (lane == 0) { sh_na = na; sh_nb = pb - na; }
    }
    __syncthreads();
    const uint32_t na = uni(sh_na), nb = uni(sh_nb), nmem = na + nb, nw = (nmem + 63) >> 6;
    for (uint32_t p = tid; p < nmem; p += SM_T) dg_l[p] = one_count<SEG, CT>(g, S.nseg, (uint32_t)S.seg_stride, (uint32_t)mpos[p] * (ld + 1));
    __syncthreads();
    double accA = 0.0, accB = 0.0, accX = 0.0;
    if (nw <= 4) hfst_small_rows<4, 4, SEG, CT>(S, tbl, g, ld, wave, lane, na, nmem, nw, mpos, dg_l, accA, accB, accX);
    else hfst_small_rows<8, 2, SEG, CT>(S, tbl, g, ld, wave, lane, na, nmem, nw, mpos, dg_l, accA, accB, accX);
    accA = block_sum(accA, shd); accB = block_sum(accB, shd); accX = block_sum(accX, shd);
    if (tid == 0) {
        const uint64_t a_ = na, b_ = nb;  // every pair is present on a Gram problem: the counts are the class sizes' products
        hfst_outputs(accA, accB, accX, a_ * (a_ - (a_ ? 1 : 0)) / 2, 0, b_ * (b_ - (b_ ? 1 : 0)) / 2, 0, a_ * b_, 0, seq_len ? seq_len[prob] : 0,
                     out + prob);
    }
}

// ---------------------------------------------------------------------------------------
// pica2.py:60-169 analyze_similarity_matrix, seeds in position order (the smallest remaining name: stats.hip greedy_groups).
// Positions 0..m are the elements idx[0..m) (nullptr: the sequences themselves), ascending.
//  Step 1 in blocks: wave 0 lists the next <= 64 free positions (the candidates, ascending); the four waves test each
//  candidate's row against the positions right of it — one ballot per 64 positions, written as one word of the candidate's
//  row — and wave 0, candidate b's row in lane b and the free set in scalar registers, replays the reference's loop:
//  candidates nobody absorbed and whose rows meet no free position become singleton groups together (they absorb nobody, and
//  nothing later looks at positions below its own seed), the first candidate with a non-empty row opens a group of itself and
//  (row AND free), the candidates it absorbed drop out, and so on.
//  Step 2-3: up to 64 groups a thread per representative pair; beyond, a wave per representative row against a
//  per-position frequency (0 where the position represents nothing), identities from the Hamming-distance memo.
template <bool SEG, typename CT>
__global__ __launch_bounds__(SM_T, SEG ? 4 : 5) void pica2_small_kernel(SimBatch batch, const uint32_t *__restrict__ idx, uint32_t m, double thr,
                                                              const uint64_t *__restrict__ seq_len, Pica2Out *__restrict__ out) {
    __shared__ double tbl[SIM_TBL_N];
    __shared__ double fpos[SM_N];
    __shared__ uint64_t rowsT[SM_W * 64];  // word k of candidate b's row at k * 64 + b
    __shared__ int32_t dg_l[SM_N];         // Gram diagonal by position
    __shared__ uint16_t epos[SM_N], rep[SM_N], gsz[SM_N], cand[64];
    __shared__ uint32_t sh_n, sh_G, sh_cursor;
    __shared__ int32_t sh_hcut;
    __shared__ double shd[SM_T / 64];
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);
    S.dense = nullptr; S.g16 = sizeof(CT) == 2 ? 1u : 0u;  // (sim_view took the problem's base from batch.g16: the launch matches CT to it)
    if (!SEG) S.nseg = 1;
    const uint32_t nseg = S.nseg, sstride = (uint32_t)S.seg_stride;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), ld = S.ld;
    const CT *__restrict__ g = reinterpret_cast<const CT *>(S.gram);
    const int32_t junk = (int32_t)lane;  // see skipped_load
    const uint32_t nw = (m + 63) >> 6;
    for (uint32_t o = tid; o < SM_N; o += SM_T) {
        const uint32_t e = o < m ? (idx ? idx[o] : o) : 0u;
        epos[o] = (uint16_t)e;
        dg_l[o] = o < m ? one_count<SEG, CT>(g, nseg, sstride, e * (ld + 1)) : 0;
    }
    if (wave == 0) {
        const int32_t hc = match_cutoff_wave(S, thr);
        if (lane == 0) { sh_hcut = hc; sh_G = 0; sh_cursor = 0; }
    }
    __syncthreads();
    const int32_t hcut = (int32_t)uni((uint32_t)sh_hcut);
    uint32_t ek[SM_W];  // element and diagonal of this lane's positions
    int32_t dgk[SM_W];
#pragma unroll
    for (uint32_t k = 0; k < SM_W; ++k) { ek[k] = epos[64 * k + lane]; dgk[k] = dg_l[64 * k + lane]; }
    uint64_t vm[SM_W];  // positions below m
#pragma unroll
    for (uint32_t k = 0; k < SM_W; ++k) vm[k] = m >= 64 * (k + 1) ? ~0ull : m > 64 * k ? ((1ull << (m - 64 * k)) - 1ull) : 0ull;
    uint64_t fr[SM_W];  // the free set (wave 0's copy is the one that counts)
#pragma unroll
    for (uint32_t k = 0; k < SM_W; ++k) fr[k] = vm[k];
    uint32_t G = 0;
    bool done = false;
    // a block resolves at least one candidate: m + 1 blocks bound the loop whatever happens (a grid must drain); running out of
    // the bound means an invariant broke — the device error word then fails the call (IMPOP_E_INTERNAL)
    for (uint32_t block = 0; block <= m; ++block) {
        const uint32_t B = block == 0 ? 16u : 64u;  // few groups are found among the first candidates: a short first block
        uint32_t cb = 0;                            // wave 0: lane b's candidate
        if (wave == 0) {
            const uint32_t cursor = uni(sh_cursor);
            uint64_t fm[SM_W];
            uint32_t pre[SM_W + 1];
            pre[0] = 0;
#pragma unroll
            for (uint32_t k = 0; k < SM_W; ++k) {
                const uint64_t msk = k < (cursor >> 6) ? 0ull : k == (cursor >> 6) ? (~0ull << (cursor & 63)) : ~0ull;
                fm[k] = fr[k] & msk;
                pre[k + 1] = pre[k] + (uint32_t)__popcll(fm[k]);
            }
            const uint32_t nfree = pre[SM_W], nb = nfree < B ? nfree : B;
            // lane b: the b-th free position at or right of the cursor
            uint32_t r = lane, wj = 0;
            uint64_t ws = fm[0];
#pragma unroll
            for (uint32_t j = 1; j < SM_W; ++j)
                if (lane >= pre[j]) { ws = fm[j]; r = lane - pre[j]; wj = j; }
            uint32_t bp = 0;
#pragma unroll
            for (uint32_t s = 32; s >= 1; s >>= 1) {
                const uint32_t c = (uint32_t)__popcll((ws >> bp) & ((1ull << s) - 1ull));
                if (r >= c) { r -= c; bp += s; }
            }
            cb = 64 * wj + (bp & 63);
            if (lane < nb) cand[lane] = (uint16_t)cb;
            if (lane == 0) sh_n = nb;
        }
        __syncthreads();
        const uint32_t nb = uni(sh_n);
        if (nb == 0) { done = true; break; }
        // the candidates' rows: two candidates (up to 16 loads) in flight per wave; what a candidate's row needs is staged in the
        // lanes first (lane j: candidate wave + 4 j) and comes back by v_readlane
        uint32_t cst = 0, rowst = 0;
        int32_t acst = 0;
        {
            const uint32_t b = wave + (SM_T / 64) * lane;
            if (b < nb) { cst = cand[b]; acst = dg_l[cst]; rowst = (uint32_t)epos[cst] * ld; }
        }
        for (uint32_t b0 = wave, j0 = 0; b0 < nb; b0 += 2 * (SM_T / 64), j0 += 2) {
            uint32_t c[2], k0[2], row[2];
            int32_t ac[2];
            bool lv[2];
            int32_t I[2][SM_W];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t b = b0 + (SM_T / 64) * u;
                lv[u] = b < nb;
                const uint32_t j = lv[u] ? j0 + u : j0;
                c[u] = (uint32_t)__builtin_amdgcn_readlane((int)cst, (int)j);
                k0[u] = c[u] >> 6;
                ac[u] = __builtin_amdgcn_readlane(acst, (int)j);
                row[u] = (uint32_t)__builtin_amdgcn_readlane((int)rowst, (int)j);
            }
            // positions ascend, so do the elements: (row, column) is in the upper triangle
            row_counts<SM_W, 2, SEG, CT>(g, nseg, sstride, lv, k0, nw, junk, [&](int u, uint32_t k) { return row[u] + ek[k]; }, I);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t b = b0 + (SM_T / 64) * u;
                if (!lv[u]) continue;
#pragma unroll
                for (uint32_t k = 0; k < SM_W; ++k) {
                    uint64_t bits = 0;
                    if (k >= k0[u] && k < nw) {
                        const int32_t H = ac[u] + dgk[k] - 2 * I[u][k];
                        const uint64_t right = k > k0[u] ? ~0ull : (c[u] & 63) == 63 ? 0ull : (~0ull << ((c[u] & 63) + 1));
                        bits = __ballot(H <= hcut) & right & vm[k];  // strict > threshold (pica2.py:106), positions right of the seed
                    }
                    if (lane == 0) rowsT[k * 64 + b] = bits;
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            uint64_t row[SM_W];
#pragma unroll
            for (uint32_t k = 0; k < SM_W; ++k) row[k] = lane < nb ? rowsT[k * 64 + lane] : 0ull;
            bool alive = lane < nb;
            uint32_t pos = 0;
            G = uni(sh_G);
            for (uint32_t it = 0; it < 65; ++it) {
                uint64_t any = 0;
#pragma unroll
                for (uint32_t k = 0; k < SM_W; ++k) any |= row[k] & fr[k];
                const bool mine = alive && lane >= pos;
                const uint64_t bal_alive = __ballot(mine), bal_busy = __ballot(mine && any != 0);
                const uint32_t first = bal_busy ? (uint32_t)__ffsll((unsigned long long)bal_busy) - 1 : 64u;
                const uint64_t lone = bal_alive & (first >= 64 ? ~0ull : ((1ull << first) - 1ull));
                if ((lone >> lane) & 1) {
                    const uint32_t gid = G + (uint32_t)__popcll(lone & ((1ull << lane) - 1ull));
                    rep[gid] = (uint16_t)cb;
                    gsz[gid] = 1;
                }
                G += (uint32_t)__popcll(lone);
                if (first >= 64) break;
                // candidate `first`: free, and its row takes free positions with it (pica2.py:100-108)
                const uint32_t cf = (uint32_t)__builtin_amdgcn_readlane((int)cb, (int)first);
                uint32_t cnt = 1;
                uint64_t mem[SM_W];
#pragma unroll
                for (uint32_t k = 0; k < SM_W; ++k) {
                    mem[k] = readlane64(row[k], first) & fr[k];
                    fr[k] &= ~mem[k];
                    cnt += (uint32_t)__popcll(mem[k]);
                }
                if (lane == 0) { rep[G] = (uint16_t)cf; gsz[G] = (uint16_t)cnt; }
                ++G;
                uint64_t mw = mem[0];  // the word of the members that holds this lane's candidate
#pragma unroll
                for (uint32_t j = 1; j < SM_W; ++j)
                    if ((cb >> 6) == j) mw = mem[j];
                if ((mw >> (cb & 63)) & 1) alive = false;
                pos = first + 1;
            }
            if (lane == 0) {
                sh_G = G;
                sh_cursor = (uint32_t)cand[nb - 1] + 1;
            }
        }
        __syncthreads();
        if (nb < B) { done = true; break; }  // the block took every free position that was left
    }
    if (!done && tid == 0 && batch.err) atomicOr(batch.err, DEV_ERR_GROUPING);
    G = uni(sh_G);
    // Step 2-3 (pica2.py:118-154): sum over pairs of groups g < h of 2 (1 - identity(rep_g, rep_h)) f_g f_h
    const double total = (double)m;
    double acc = 0.0;
    if (G > 64) {
        for (uint32_t h = tid; h < SIM_TBL_N; h += SM_T) tbl[h] = match_identity(S.W, (int64_t)h, S.round_digits);
        for (uint32_t o = tid; o < SM_N; o += SM_T) fpos[o] = 0.0;
        __syncthreads();
        for (uint32_t gi = tid; gi < G; gi += SM_T) fpos[rep[gi]] = (double)gsz[gi] / total;
        __syncthreads();
        double fj[SM_W];
#pragma unroll
        for (uint32_t k = 0; k < SM_W; ++k) fj[k] = fpos[64 * k + lane];
        // rows as in hfst_small_kernel: staged in the lanes (lane j of set t: group wave + 4 (j + 64 t)), two in flight
        uint32_t rrs[2], rows_[2];
        int32_t ars[2];
        double fis[2];
#pragma unroll
        for (uint32_t t = 0; t < 2; ++t) {
            const uint32_t gi = wave + (SM_T / 64) * (lane + 64 * t);
            rrs[t] = gi < G ? (uint32_t)rep[gi] : 0u;
            ars[t] = dg_l[rrs[t]];
            rows_[t] = (uint32_t)epos[rrs[t]] * ld;
            fis[t] = gi < G ? fpos[rrs[t]] : 0.0;
        }
        const uint32_t nrows = G > wave ? (G - wave + SM_T / 64 - 1) / (SM_T / 64) : 0;
        for (uint32_t r0 = 0; r0 < nrows; r0 += 2) {
            uint32_t rr[2], k0[2], row[2];
            int32_t ar[2];
            bool lv[2];
            int32_t I[2][SM_W];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lv[u] = r0 + u < nrows;
                const uint32_t r = lv[u] ? r0 + u : r0;
                const uint32_t t = r >> 6, j = r & 63;
                rr[u] = (uint32_t)__builtin_amdgcn_readlane((int)(t ? rrs[1] : rrs[0]), (int)j);
                ar[u] = __builtin_amdgcn_readlane(t ? ars[1] : ars[0], (int)j);
                row[u] = (uint32_t)__builtin_amdgcn_readlane((int)(t ? rows_[1] : rows_[0]), (int)j);
                k0[u] = rr[u] >> 6;
            }
            row_counts<SM_W, 2, SEG, CT>(g, nseg, sstride, lv, k0, nw, junk, [&](int u, uint32_t k) { return row[u] + ek[k]; }, I);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!lv[u]) continue;
                double racc = 0.0;
#pragma unroll
                for (uint32_t k = 0; k < SM_W; ++k) {
                    if (k >= k0[u] && k < nw) {
                        const uint32_t o = 64 * k + lane;
                        const int32_t H = (o > rr[u] && o < m) ? ar[u] + dgk[k] - 2 * I[u][k] : 0;  // no pair here: distance 0, identity 1, term 0
                        racc += (1 - ident_of(S, tbl, H)) * fj[k];
                    }
                }
                // the row's own frequency: lane (row & 63) of its staging set holds it; every lane takes that lane's value
                const uint32_t r = r0 + u;
                const double fsel = (r >> 6) ? fis[1] : fis[0];
                const double fi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fsel), (int)(r & 63)),
                                                   __builtin_amdgcn_readlane(__double2loint(fsel), (int)(r & 63)));
                acc += fi * racc;
            }
        }
    } else {
        for (uint32_t t = tid; t < G * G; t += SM_T) {
            const uint32_t a = t / G, b = t - a * G;
            if (a >= b) continue;
            const uint32_t ra = rep[a], rb = rep[b];  // ra < rb: the groups are numbered by their seeds, which ascend
            const int32_t H = dg_l[ra] + dg_l[rb] - 2 * one_count<SEG, CT>(g, nseg, sstride, (uint32_t)epos[ra] * ld + epos[rb]);
            const double s = match_identity(S.W, (int64_t)H, S.round_digits);
            acc += ((1 - s) * ((double)gsz[a] / total)) * ((double)gsz[b] / total);
        }
    }
    acc = 2 * block_sum(acc, shd);
    if (tid == 0) {
        double pi = 0.0, pi_site = 0.0;
        if (m != 0 && G > 1) {  // every pair is present on a Gram problem
            pi = ((double)m / (double)(m - 1)) * acc;  // pica2.py:154
            const uint64_t L = seq_len ? seq_len[prob] : 0;
            pi_site = L ? pi / (double)L : __builtin_nan("");  // :163-164, None -> NaN
        }
        Pica2Out o;
        o.pi = pi; o.pi_site = pi_site; o.n_groups = G; o.pad = 0;
        o.sum_2pairs = acc; o.n_pairs = G > 1 ? (uint64_t)G * (G - 1) / 2 : 0;
        out[prob] = o;
    }
}

bool small_shape(const SimBatch &b) {
    static const bool off = [] { const char *e = getenv("IMPOP_EPILOGUE_SMALL"); return e && e[0] == '0'; }();  // A/B and test switch
    // counts of 16 or 32 bits; Hamming distances are formed in 32-bit arithmetic: W < 2^30 (uint16 counts imply W < 2^16)
    const bool w_ok = b.g16 || (b.max_W != 0 && b.max_W < (1ull << 30));
    return !off && b.gram && !b.dense && w_ok && (b.seg_first != nullptr) == (b.seg_count != nullptr) && b.kind == IMPOP_IDENTITY_MATCH &&
           b.ld <= 4096;  // (element offsets inside a matrix are 32-bit, elements and positions 16-bit)
}

}  // namespace

bool pica2_small_applies(const SimBatch &b, uint32_t n_el, const uint32_t *d_order, const uint32_t *d_group_of) {
    return small_shape(b) && n_el <= SM_N && !d_order && !d_group_of;
}
bool hfst_small_applies(const SimBatch &b) { return small_shape(b) && b.n <= SM_N; }

int launch_pica2_small(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_idx, uint32_t n_el, double threshold,
                       const uint64_t *d_seq_len, Pica2Out *d_out) {
    SimBatch be = b;
    be.err = ctx->d_err;
#define LAUNCH_P2(SEG, CT)                                                                                                      \
    hipLaunchKernelGGL((pica2_small_kernel<SEG, CT>), dim3((uint32_t)n_problems), dim3(SM_T), 0, ctx->stream, be, d_idx, n_el, threshold, \
                       d_seq_len, d_out)
    if (b.seg_first) { if (b.g16) LAUNCH_P2(true, uint16_t); else LAUNCH_P2(true, int32_t); }
    else { if (b.g16) LAUNCH_P2(false, uint16_t); else LAUNCH_P2(false, int32_t); }
#undef LAUNCH_P2
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

int launch_hfst_small(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint8_t *d_in_a, const uint8_t *d_in_b,
                      const uint64_t *d_seq_len, HfstOut *d_out) {
#define LAUNCH_HF(SEG, CT) \
    hipLaunchKernelGGL((hfst_small_kernel<SEG, CT>), dim3((uint32_t)n_problems), dim3(SM_T), 0, ctx->stream, b, d_in_a, d_in_b, d_seq_len, d_out)
    if (b.seg_first) { if (b.g16) LAUNCH_HF(true, uint16_t); else LAUNCH_HF(true, int32_t); }
    else { if (b.g16) LAUNCH_HF(false, uint16_t); else LAUNCH_HF(false, int32_t); }
#undef LAUNCH_HF
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

}  // namespace impop
