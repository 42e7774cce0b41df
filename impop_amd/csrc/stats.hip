// stats.hip — pica2 / h-fst / af / tj_d arithmetic on the GPU + their C-ABI entry points for
// a caller-supplied identity matrix (the .sim drop-in path).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "stats_kernels.h"

namespace impop {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int ST = 256;  // threads per problem

// the wave's index in its workgroup AS A SCALAR: the compiler takes threadIdx.x >> 6 for a per-lane value, and every loop and
// branch on it for divergent (execution masks, loads parked behind their own waits) unless it is told
__device__ __forceinline__ uint32_t wave_index() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// deterministic block sums (wave butterfly, then waves in order)
__device__ __forceinline__ double block_sum_f64(double v, double *sh /*>= ST/64*/) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < ST / 64; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t *sh) {
    v = wave_sum_u64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t t = 0;
    for (int w = 0; w < ST / 64; ++w) t += sh[w];
    return t;
}

// ---------------------------------------------------------------------------------------
// Greedy grouping shared by pica2 (pica2.py:94-112) and hud.py (hud.py:64-86): repeatedly take a seed from
// the remaining elements and move every remaining element whose identity to the SEED exceeds the threshold
// (strict >) into the seed's group.  The reference takes its seeds with set.pop(), i.e. in the iteration
// order of `set(elements)`; `order` (nullable) is that order as element positions (order[k] = position in
// 0..m of the k-th element the set iterates), nullptr = positions 0, 1, 2, ... (seed = smallest remaining
// name).  Groups come out numbered the way the reference numbers them after `sorted(group)` / `groups.sort()`:
// by their smallest member; rep[g] (nullable) = that member's position, gsz[g] = group size.
// scratch: 2*m u32 of LDS the caller can spare during grouping (only touched when order != nullptr).
// All threads of the workgroup call this; returns the number of groups.
// `match` identity of a Gram problem: round((W - H) / W) is non-increasing in the Hamming distance H, so "identity >
// threshold" is "H <= H*" for one H* per problem (-1: never), found by bisection with the very function the pairs
// would be tested with — the same decisions, three integer operations per pair instead of a division and a decimal
// rounding.  Every thread computes the same value (about 32 evaluations).
__device__ __forceinline__ int64_t match_cutoff(const SimView &S, double thr) {
    const int64_t W = (int64_t)S.W;
    auto above = [&](int64_t H) { const double v = match_identity(S.W, H, S.round_digits); return v == v && v > thr; };
    if (!above(0)) return -1;
    int64_t lo = 0, hi = W > 0 ? W : 0;  // above(lo) holds
    if (above(hi)) return hi;
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (above(mid)) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------
// Greedy grouping on bits (seeds in position order).  Whether o joins seed i does not depend on the state of the
// grouping — only whether o is still free does.  So the join tests of a BLOCK of candidate seeds (the next free
// positions) are made together, one bit each, and the reference's loop (pica2.py:94-112) then runs on those bits in one
// wave, without a barrier or a memory round trip per group (the serial loop of greedy_groups pays both: 2.5 us per
// group at 465 elements, 6 us at 4096 — 25 ms for 4096 singleton groups, 100 of the 144 ms of config 5).
// Seeds ascend, so everything below a seed is already grouped and only o > seed is tested: the upper triangle, the part
// the Gram kernel writes, read along rows.
// Bit layout: position o is bit (o % 256) / 4 of word 4 (o / 256) + o % 4.  A wave covers 256 consecutive positions of a
// row with ONE 16-byte load per lane (lane L: positions 4L .. 4L+3) and its four ballots ARE the four words of that
// 256-block.  aw = 4 ceil(m / 256) words per row.
// Where the bits come from:
//  * FROM_ADJ: pica2_adj_kernel (grid: problems x row chunks) has tested ALL pairs o > i of a large problem up front,
//    over the whole chip; a block is up to 64 rows of that matrix, fetched in one batch;
//  * otherwise the workgroup tests the block's pairs itself — (candidate, 256-block) items dealt to the waves, JU in
//    flight per lane — with blocks just long enough to fill two such batches per wave.
// Measured (tools/time_epilogue.py, 4096 x 465-haplotype 10 kb windows; tools/bench_config5.py): config 5 (4096
// haplotypes, 200 x 50 kb windows, singleton-heavy) 144 -> 53 ms end to end; at 465 haplotypes the grouping of 465
// singleton groups 985 -> 600 us per problem (pica2_kernel 6.9 -> 5.0 ms per 4096 windows), 8 groups 38 -> 32 us.  What is
// left at 465 is wave 0's share of a SIMD it shares with four other problems' waves, not barriers or round trips.
// Then wave 0, the free set in its registers (lane l holds words l and l + 64), walks the block's candidates in order: a
// candidate an earlier seed of the block absorbed is skipped, any other opens a group of itself plus (its row AND the
// free set).  All threads call; returns the number of groups; m <= 8192.
struct Pica2Split {
    uint32_t *tab;     // problem p: rep[n_el] | gsz[n_el] | G | have | npairs lo | npairs hi
    double *rowsum;    // problem p: n_el doubles
    uint64_t *adj;     // nullable; problem p: n_el rows of 4 ceil(n_el / 256) words (layout above): "o > i and o joins seed i"
};
__device__ __forceinline__ uint32_t *split_tab(const Pica2Split &sp, uint64_t p, uint32_t n_el) { return sp.tab + p * (2ull * n_el + 4); }

__host__ __device__ __forceinline__ uint32_t bit_words(uint32_t m) { return 4 * ((m + 255) / 256); }
__device__ __forceinline__ uint32_t bit_word(uint32_t o) { return 4 * (o >> 8) + (o & 3); }
__device__ __forceinline__ uint32_t bit_lane(uint32_t o) { return (o & 255) >> 2; }
__device__ __forceinline__ uint32_t bit_pos(uint32_t w, uint32_t L) { return 256 * (w >> 2) + 4 * L + (w & 3); }
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l /*uniform*/) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}
// word w (< 128, PER LANE) of a 128-word set whose words l and l + 64 live in lane l's u0 / u1: a register exchange
// (ds_bpermute_b32: lane i receives the source register of lane addr_i / 4).  Every lane of the wave must execute it.
__device__ __forceinline__ uint64_t lane_word(uint64_t u0, uint64_t u1, uint32_t w) {
    const int a = (int)((w & 63) << 2);
    const uint32_t lo0 = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)u0);
    const uint32_t hi0 = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)(u0 >> 32));
    const uint32_t lo1 = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)u1);
    const uint32_t hi1 = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)(u1 >> 32));
    return w < 64 ? (((uint64_t)hi0 << 32) | lo0) : (((uint64_t)hi1 << 32) | lo1);
}

struct JoinTest {  // "identity(seed, o) > threshold", as greedy_groups tests it
    bool by_cutoff;  // Gram + match: H <= H* (match_cutoff)
    bool vec;        // Gram problem whose rows can be read as aligned quads at the positions themselves
    int64_t hstar;
    double thr;
};
__device__ __forceinline__ JoinTest join_test(const SimView &S, const uint32_t *idx, double thr) {
    JoinTest t;
    t.by_cutoff = S.gram && S.kind == IMPOP_IDENTITY_MATCH;
    t.hstar = t.by_cutoff ? match_cutoff(S, thr) : 0;
    t.thr = thr;
    t.vec = S.gram && !idx && gram_quads_aligned(S);
    return t;
}
typedef int i32q __attribute__((ext_vector_type(4)));
// word `k` (0..3, varying by lane) of four uniform words, without indexing a register array by a lane value
__device__ __forceinline__ uint64_t pick4(const uint64_t (&w)[4], uint32_t k) { return k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3]; }

// the four words of 256-block hs[k] of candidate cs[k]'s row, for U items (live[k] false = nothing to do): all loads of
// the U items go out before the first is consumed.  Registers decide how many problems a CU holds at once, so only what
// must wait for memory is kept per item: the shared counts I (the two diagonal entries come from LDS when the bit is
// formed) or the identities.
template <int U>
__device__ __forceinline__ void join_blocks(const SimView &S, const JoinTest &T, const uint32_t *__restrict__ idx, uint32_t m,
                                   const uint32_t (&cs)[U], const uint32_t (&hs)[U], const bool (&live)[U], uint64_t (&bits)[U][4]) {
    const uint32_t lane = threadIdx.x & 63;
    if (S.gram) {
        i32q iv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t o = 256 * hs[u] + 4 * lane;
            iv[u] = i32q{0, 0, 0, 0};
            if (!live[u] || o + 3 <= cs[u] || o >= m) continue;  // no position of the quad is above the seed and in range
            if (T.vec) {
                const uint64_t g = (uint64_t)cs[u] * S.ld + o;  // o + 3 < ld: ld is a multiple of 4 and o < m <= ld
                // (a window without a single Gram segment — no sites, or on a compacted matrix no VARIABLE site, among overlapping
                // windows — has nseg == 0: its counts are the constant alone, not the first segment of the chunk)
                i32q v = i32q{0, 0, 0, 0};
                for (uint32_t k = 0; k < S.nseg; ++k) v += gram_ld4(S, g + k * S.seg_stride);
                iv[u] = v + (int32_t)S.add;
            } else {
                const uint32_t es = idx ? idx[cs[u]] : cs[u];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o + j > cs[u] && o + j < m) iv[u][j] = (int32_t)gram_at(S, es, idx ? idx[o + j] : o + j);  // positions ascend, so do the elements
            }
        }
        if (T.vec && T.by_cutoff && S.diag && S.W < (1ull << 30)) {
            // the common shape — rows read as quads at the positions themselves (no element list), `match` identity decided by the
            // integer cutoff, diagonal in LDS, counts below 2^30: 32-bit arithmetic, the four diagonal entries of a quad as one
            // 16-byte LDS read, the compare IS the ballot.  ~8 instructions per position instead of ~30 (64-bit Hamming arithmetic,
            // a scalar LDS read and an element-list test per position): with hundreds of groups per window these tests were 1.3 ms of
            // pica2's 3.1 ms per 4096 windows (profiles/r03_epilogue_kernels.txt)
            const int32_t hcut = T.hstar > 0x7FFFFFFFll ? 0x7FFFFFFF : T.hstar < -1 ? -1 : (int32_t)T.hstar;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t o0 = 256 * hs[u] + 4 * lane;
                const int32_t as = live[u] ? S.diag[cs[u] < m ? cs[u] : 0] : 0;
                i32q dg = i32q{0, 0, 0, 0};
                if (live[u] && o0 + 3 < m) dg = *reinterpret_cast<const i32q *>(S.diag + o0);  // o0 is a multiple of 4: aligned
                else if (live[u]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dg[j] = o0 + j < m ? S.diag[o0 + j] : 0;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t o = o0 + j;
                    const int32_t H = as + dg[j] - 2 * iv[u][j];
                    bits[u][j] = __ballot(live[u] && o > cs[u] && o < m && H <= hcut);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t es = idx ? idx[cs[u] < m ? cs[u] : 0] : cs[u];
            const int64_t as = live[u] ? (S.diag ? (int64_t)S.diag[es] : gram_at(S, es, es)) : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t o = 256 * hs[u] + 4 * lane + j;
                bool joins = false;
                if (live[u] && o > cs[u] && o < m) {
                    const uint32_t eo = idx ? idx[o] : o;
                    const int64_t ao = S.diag ? (int64_t)S.diag[eo] : gram_at(S, eo, eo);
                    if (T.by_cutoff) {
                        joins = as + ao - 2 * (int64_t)iv[u][j] <= T.hstar;
                    } else {
                        const double v = sim_from_gram(S, (int64_t)iv[u][j], as, ao);
                        joins = v == v && v > T.thr;
                    }
                }
                bits[u][j] = __ballot(joins);
            }
        }
    } else {
        // dense identities (the .sim path: one problem per call): eight loads in flight are enough, two items at a time
        static_assert(U % 2 == 0, "items in pairs");
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += 2) {
            double sv[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t o = 256 * hs[u0 + k] + 4 * lane + j;
                    sv[k][j] = 0.0;
                    if (live[u0 + k] && o > cs[u0 + k] && o < m) sv[k][j] = sim_get(S, idx ? idx[cs[u0 + k]] : cs[u0 + k], idx ? idx[o] : o);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t o = 256 * hs[u0 + k] + 4 * lane + j;
                    const bool in = live[u0 + k] && o > cs[u0 + k] && o < m;
                    bits[u0 + k][j] = __ballot(in && sv[k][j] == sv[k][j] && sv[k][j] > T.thr);  // strict > (pica2.py:106, hud.py:80)
                }
            }
        }
    }
}

// dynamic LDS: diag[batch.n] int32 (Gram problems)
__global__ __launch_bounds__(ST) void pica2_adj_kernel(SimBatch batch, const uint32_t *__restrict__ idx, uint32_t n_el,
                                                       double threshold, Pica2Split split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (S.gram) {
        int32_t *diag_l = reinterpret_cast<int32_t *>(lds_raw);
        for (uint32_t i = tid; i < batch.n; i += ST) diag_l[i] = (int32_t)gram_at(S, i, i);
        S.diag = diag_l;
    }
    const JoinTest T = join_test(S, idx, threshold);
    __syncthreads();
    const uint32_t aw = bit_words(n_el), nh = aw / 4;
    uint64_t *adj = split.adj + prob * (uint64_t)n_el * aw;
    constexpr int AU = 4;
    for (uint32_t i = wave_index() + (ST / 64) * blockIdx.y; i < n_el; i += (ST / 64) * gridDim.y) {
        for (uint32_t h0 = i >> 8; h0 < nh; h0 += AU) {
            uint32_t cs[AU], hs[AU];
            bool live[AU];
            uint64_t bits[AU][4];
#pragma unroll
            for (int u = 0; u < AU; ++u) { cs[u] = i; hs[u] = h0 + u; live[u] = h0 + u < nh; }
            join_blocks<AU>(S, T, idx, n_el, cs, hs, live, bits);
#pragma unroll
            for (int u = 0; u < AU; ++u)
                if (lane < 4 && live[u]) adj[(uint64_t)i * aw + 4 * (h0 + u) + lane] = pick4(bits[u], lane);
        }
    }
}

constexpr uint32_t GG_ROWS = 512;  // words of candidate rows a small problem's block may take (the kernels declare them)
template <bool FROM_ADJ>
__device__ __forceinline__ uint32_t greedy_groups_bits(const uint64_t *__restrict__ adj, const SimView &S, const uint32_t *__restrict__ idx,
                                              double thr, uint32_t m, uint32_t *grp, uint32_t *gsz, uint32_t *rep, uint64_t *rows_big,
                                              uint64_t *rows_small /*LDS, GG_ROWS words (unused with FROM_ADJ)*/) {
    constexpr int JU = 4;               // (candidate, 256-block) items in flight per wave
    constexpr uint32_t ROWS_S = GG_ROWS;
    __shared__ uint32_t cand[64];
    __shared__ uint32_t sh_n, sh_G, sh_B;
    if (m == 0) return 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t aw = bit_words(m), nh = aw / 4;
    uint64_t *rows = FROM_ADJ ? rows_big : rows_small;  // rows_big: m words (the caller's scratch)
    // B = candidates per block.  It starts small (few groups: most candidates of a block are absorbed by an earlier seed of the
    // same block, their join tests wasted) and DOUBLES, up to what the row words hold, after every block in which at least half the
    // candidates opened a group — with hundreds of groups per window the blocks' fixed costs (two barriers, a round of memory
    // latency, the serial resolution) were what the grouping's time went on: 29 blocks of 16 at 465 singleton groups, 9 when grown
    uint32_t B, Bcap;
    if (FROM_ADJ) {
        B = Bcap = m / aw < 64 ? m / aw : 64;
    } else {
        Bcap = ROWS_S / aw < 64 ? ROWS_S / aw : 64;                    // >= 4: aw <= 128
        const uint32_t fill = (2 * (ST / 64) * JU + nh - 1) / nh;      // candidates whose blocks make two batches per wave
        B = fill < 4 ? 4 : fill;
        if (B > Bcap) B = Bcap;
    }
    const JoinTest T = FROM_ADJ ? JoinTest{false, false, 0, 0.0} : join_test(S, idx, thr);
    for (uint32_t i = tid; i < m; i += ST) { grp[i] = NONE; gsz[i] = 0; }
    auto initial = [&](uint32_t w) -> uint64_t {  // bit L of word w: position bit_pos(w, L) < m
        const uint32_t base = 256 * (w >> 2) + (w & 3);
        if (w >= aw || m <= base) return 0;
        const uint32_t c = (m - base + 3) / 4;
        return c >= 64 ? ~0ull : ((1ull << c) - 1);
    };
    uint64_t u0 = initial(lane), u1 = initial(lane + 64);  // the free set (wave 0's copy is the one that counts)
    uint32_t G = 0, h_first = 0;
    if (tid == 0) { sh_G = 0; sh_B = B; }
    __syncthreads();
    // every block resolves at least one candidate: m blocks bound the loop whatever happens (a grid must drain).  Running
    // out of the bound means an invariant of this function broke: the device error word makes the CALL fail
    // (IMPOP_E_INTERNAL) instead of handing back partial groups.
    bool done = false;
    for (uint32_t guard = 0; guard <= m; ++guard) {
        uint32_t mycand = 0;  // wave 0: lane b holds candidate b
        if (tid < 64) {  // the next up-to-B free positions, ascending: uniform work on words read across the lanes
            // the serial stretches of one wave: ahead of the other problems' waves on this SIMD while they last
            __builtin_amdgcn_s_setprio(3);
            uint32_t n = 0;
            bool first = true;
            // (h_first lives across the workgroup-wide loop in a vector register: tell the compiler it is one value per wave,
            // or every readlane below turns into a loop over lanes)
            for (uint32_t h = __builtin_amdgcn_readfirstlane(h_first); h < nh && n < B; ++h) {
                uint64_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = readlane_u64(4 * h + j < 64 ? u0 : u1, (4 * h + j) & 63);
                uint64_t any = w[0] | w[1] | w[2] | w[3];
                if (any && first) { h_first = h; first = false; }
                while (any && n < B) {
                    const uint32_t L = (uint32_t)__ffsll((unsigned long long)any) - 1;
                    any &= any - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((w[j] >> L) & 1) && n < B) {
                            if (lane == n) mycand = 256 * h + 4 * L + j;
                            ++n;
                        }
                }
            }
            if (lane < n) cand[lane] = mycand;
            if (lane == 0) sh_n = n;
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const uint32_t n = __builtin_amdgcn_readfirstlane(sh_n);
        if (n == 0) { done = true; break; }
        if (FROM_ADJ) {
            constexpr int RU = 8;  // candidate rows in flight per wave
            for (uint32_t b0 = wave_index(); b0 < n; b0 += (ST / 64) * RU) {
                uint64_t v[RU][2];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const uint32_t b = b0 + (ST / 64) * u;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t w = lane + 64 * hh;
                        v[u][hh] = 0;
                        if (b < n && w < aw && w >= 4 * (cand[b] >> 8)) v[u][hh] = adj[(uint64_t)cand[b] * aw + w];
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const uint32_t b = b0 + (ST / 64) * u;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t w = lane + 64 * hh;
                        if (b < n && w < aw) rows[(uint64_t)b * aw + w] = v[u][hh];
                    }
                }
            }
        } else {
            const uint32_t items = n * nh;  // item = (candidate b, block h): rows[b * aw + 4 h .. + 4]
            for (uint32_t it0 = wave_index() * JU; it0 < items; it0 += (ST / 64) * JU) {
                uint32_t cs[JU], hs[JU];
                bool live[JU];
                uint64_t bits[JU][4];
#pragma unroll
                for (int u = 0; u < JU; ++u) {
                    const uint32_t it = it0 + u;
                    live[u] = it < items;
                    const uint32_t b = live[u] ? it / nh : 0;
                    cs[u] = cand[b];
                    hs[u] = it - b * nh;
                }
#if defined(IMPOP_PICA2_ABLATE) && (IMPOP_PICA2_ABLATE & 2)  // timing-only build: no join tests (every candidate a singleton)
#pragma unroll
                for (int u = 0; u < JU; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bits[u][j] = 0;
#else
                join_blocks<JU>(S, T, idx, m, cs, hs, live, bits);
#endif
#pragma unroll
                for (int u = 0; u < JU; ++u)
                    if (lane < 4 && live[u]) rows[4 * (it0 + u) + lane] = pick4(bits[u], lane);
            }
        }
        __syncthreads();
        if (tid < 64) {
            __builtin_amdgcn_s_setprio(3);
            // Lane b looks at candidate b: is it still free, and does its row meet the free set at all?  Candidates that are
            // free and whose rows are empty — up to the first candidate with a non-empty row — cannot be absorbed by anybody
            // in between and absorb nobody: they become singleton groups TOGETHER (ranks from a ballot).  Then the first
            // non-empty candidate is resolved the reference's way, the flags are taken again (the free set changed), and so
            // on.  With hundreds of groups per window most candidates are such singletons: one pass instead of a dependent
            // chain of LDS reads, lane reads and bit loops per candidate.
            // The free set lives in the lanes (lane l: words l and l + 64) and the look-ups are lane-indexed reads of OTHER
            // lanes' registers: ds_bpermute for the per-lane word of "is my candidate still free", v_readlane (uniform
            // index) for the walk over the words of the set.  No memory in between — the first version passed the set
            // through LDS, where the compiler may forward a lane's own earlier store over what other lanes wrote since
            // (groups were counted twice; found by tools/soak_groups.py, pinned by test_grouping_soak_shape_regression).
            uint32_t pos = 0;
            while (pos < n) {
                const uint32_t c = mycand;
                const bool mine = lane >= pos && lane < n;
                // every lane of the wave executes the exchange (a bpermute reads the registers of active lanes only)
                const uint64_t fw = lane_word(u0, u1, mine ? bit_word(c) : 0);
                const bool alive = mine && ((fw >> bit_lane(c)) & 1);
                bool nonempty = false;
                const uint32_t w_first = 4 * (c >> 8);
                for (uint32_t w = 0; w < aw; ++w) {  // uniform loop: word w of the free set is one scalar for the wave
                    const uint64_t f = readlane_u64(w < 64 ? u0 : u1, w & 63);
                    if (f == 0) continue;
                    if (alive && w >= w_first && (rows[(uint64_t)lane * aw + w] & f)) nonempty = true;
                }
                const uint64_t bal_alive = __ballot(alive), bal_busy = __ballot(nonempty);
                const uint32_t first = bal_busy ? (uint32_t)__ffsll((unsigned long long)bal_busy) - 1 : n;  // >= pos
                uint64_t lone = bal_alive & (first >= 64 ? ~0ull : ((1ull << first) - 1));                // free, row empty, before `first`
                if ((lone >> lane) & 1) {
                    const uint32_t gid = G + (uint32_t)__popcll(lone & ((1ull << lane) - 1));
                    grp[c] = gid;
                    gsz[gid] = 1;
                    if (rep) rep[gid] = c;
                }
                G += (uint32_t)__popcll(lone);
                while (lone) {  // their bits leave the free set: uniform walk, the holder lane clears
                    const uint32_t bsel = (uint32_t)__ffsll((unsigned long long)lone) - 1;
                    lone &= lone - 1;
                    const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)mycand, (int)bsel);
                    const uint32_t sw = bit_word(cs);
                    if (lane == (sw & 63)) { if (sw < 64) u0 &= ~(1ull << bit_lane(cs)); else u1 &= ~(1ull << bit_lane(cs)); }
                }
                if (first >= n) break;
                {   // candidate `first`: free, and its row takes free elements with it (pica2.py:100-108)
                    const uint32_t cf = (uint32_t)__builtin_amdgcn_readlane((int)mycand, (int)first);
                    const uint32_t cw = bit_word(cf), cl = bit_lane(cf), hl = cw & 63;
                    uint64_t n0 = (lane < aw ? rows[(uint64_t)first * aw + lane] : 0) & u0;
                    uint64_t n1 = (lane + 64 < aw ? rows[(uint64_t)first * aw + lane + 64] : 0) & u1;
                    if (lane == hl) { if (cw < 64) n0 |= 1ull << cl; else n1 |= 1ull << cl; }  // the seed itself
                    u0 &= ~n0; u1 &= ~n1;
                    const uint32_t cnt = (uint32_t)__popcll(n0) + (uint32_t)__popcll(n1);
                    if (cnt) atomicAdd(&gsz[G], cnt);
                    while (n0) { grp[bit_pos(lane, (uint32_t)__ffsll((unsigned long long)n0) - 1)] = G; n0 &= n0 - 1; }
                    while (n1) { grp[bit_pos(lane + 64, (uint32_t)__ffsll((unsigned long long)n1) - 1)] = G; n1 &= n1 - 1; }
                    if (lane == 0 && rep) rep[G] = cf;
                    ++G;
                }
                pos = first + 1;
            }
            if (lane == 0) {
                const uint32_t opened = G - sh_G;  // groups this block opened (sh_G still holds the count before it)
                sh_B = (2 * opened >= n && 2 * B <= Bcap) ? 2 * B : B;
                sh_G = G;
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        G = __builtin_amdgcn_readfirstlane(sh_G);
        B = __builtin_amdgcn_readfirstlane(sh_B);
    }
    if (!done && tid == 0 && S.err) atomicOr(S.err, DEV_ERR_GROUPING);
    return G;
}

__device__ __forceinline__ uint32_t greedy_groups(const SimView &S, const uint32_t *__restrict__ idx, uint32_t m, double thr,
                                         const uint32_t *__restrict__ order, uint32_t *grp, uint32_t *gsz, uint32_t *rep,
                                         uint32_t *scratch, uint64_t *rows_s /*LDS, GG_ROWS words*/) {
    // seeds in position order: the blocked form above (same groups, no barrier and round trip per group); the loop below
    // serves a handed-in seed order, where a seed's group may reach below it
    if (!order && m <= 8192) return greedy_groups_bits<false>(nullptr, S, idx, thr, m, grp, gsz, rep, reinterpret_cast<uint64_t *>(scratch), rows_s);
    const bool by_cutoff = S.gram && S.kind == IMPOP_IDENTITY_MATCH;
    const int64_t hstar = by_cutoff ? match_cutoff(S, thr) : 0;
    __shared__ uint32_t chunk_cnt[ST];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < m; i += ST) { grp[i] = NONE; gsz[i] = 0; }
    __syncthreads();
    uint32_t G = 0;
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t seed = order ? order[k] : k;
        if (grp[seed] != NONE) continue;  // uniform across the workgroup
        const uint32_t es = idx ? idx[seed] : seed;
        uint32_t cnt = 0;
        // without an order everything below the seed is already grouped.  Candidates in batches of GG_U per thread: all
        // Gram / table loads of a batch go out before the first is consumed (one at a time, a step of this serial loop
        // cost a full memory latency per candidate: 15 us per group at 4096 elements)
        constexpr int GG_U = 8;
        const int64_t a_seed = by_cutoff ? (S.diag ? (int64_t)S.diag[es] : gram_at(S, es, es)) : 0;
        for (uint32_t o0 = (order ? 0 : seed + 1) + tid; o0 < m; o0 += ST * GG_U) {
            int64_t hv[GG_U];
            double sv[GG_U];
#pragma unroll
            for (int u = 0; u < GG_U; ++u) {
                const uint32_t o = o0 + u * ST;
                hv[u] = 0; sv[u] = 0.0;
                if (o >= m || o == seed) continue;
                const uint32_t eo = idx ? idx[o] : o;
                if (by_cutoff) {
                    const int64_t aj = S.diag ? (int64_t)S.diag[eo] : gram_at(S, eo, eo);
                    hv[u] = a_seed + aj - 2 * gram_at(S, es < eo ? es : eo, es < eo ? eo : es);
                } else {
                    sv[u] = sim_get(S, es, eo);
                }
            }
#pragma unroll
            for (int u = 0; u < GG_U; ++u) {
                const uint32_t o = o0 + u * ST;
                if (o >= m || o == seed || grp[o] != NONE) continue;
                const bool joins = by_cutoff ? hv[u] <= hstar : (sv[u] == sv[u] && sv[u] > thr);  // strict > (pica2.py:106, hud.py:80)
                if (joins) { grp[o] = G; ++cnt; }
            }
        }
        __syncthreads();  // every thread has tested grp[seed] and its own candidates before the seed is marked
        if (tid == 0) { grp[seed] = G; ++cnt; if (rep) rep[G] = seed; }
        if (cnt) atomicAdd(&gsz[G], cnt);
        ++G;
        __syncthreads();
    }
    if (!order) return G;  // seeds were the groups' smallest members, taken in increasing order: already sorted
    // renumber by smallest member (pica2.py:110-112: sorted(group), groups.sort())
    uint32_t *gmin = scratch, *newid = scratch + m;
    for (uint32_t g = tid; g < G; g += ST) gmin[g] = NONE;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += ST) atomicMin(&gmin[grp[i]], i);
    __syncthreads();
    // new id of a group = number of group minima at smaller positions: chunked count + scan over the threads
    const uint32_t per = (m + ST - 1) / ST, lo = tid * per, hi = (lo + per < m) ? lo + per : m;
    uint32_t c = 0;
    for (uint32_t i = lo; i < hi; ++i) c += gmin[grp[i]] == i;
    chunk_cnt[tid] = c;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t t = 0; t < tid; ++t) base += chunk_cnt[t];
    for (uint32_t i = lo; i < hi; ++i)
        if (gmin[grp[i]] == i) newid[grp[i]] = base++;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += ST) grp[i] = newid[grp[i]];
    for (uint32_t g = tid; g < G; g += ST) gsz[g] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += ST) atomicAdd(&gsz[grp[i]], 1u);
    if (rep)
        for (uint32_t g = tid; g < G; g += ST) rep[newid[g]] = gmin[g];
    __syncthreads();
    return G;
}

// ---------------------------------------------------------------------------------------
// pica2.analyze_similarity_matrix (pica2.py:60-169).  Elements are idx[0..n_el) (or 0..n_el
// when idx == nullptr), already in lexicographic name order; `order`: see greedy_groups.
// dynamic LDS: rowsum[n_el] f64 (grouping scratch before step 2) | grp[n_el] u32 | rep[n_el] u32 | gsz[n_el] u32
// Large problems (>= 1024 elements, few of them): the G(G-1)/2 representative pairs of Step 2 are more work than one
// workgroup should do alone (4096 singleton groups: 8.4 M identities, 45 ms on four waves).  pica2_kernel then stops
// after the grouping and leaves rep / group size / G in `split` (per problem: 2 n_el + 4 dwords, then n_el doubles of
// row sums); pica2_rows_kernel (grid: problems x row chunks) fills the row sums; pica2_finish_kernel adds them in row
// order and writes the record.  Small problems do everything in pica2_kernel, as before.

// FAST: what the launch knows about a batch of Gram problems, as compile-time constants — 0 nothing (any batch), 1 uint16 counts
// and exactly one Gram matrix per problem, 2 int32 counts and one matrix.  With them the "16 or 32 bit" branch and the segment
// loop around every Gram load fold away, and the loads of a batch of items really are in flight together (as runtime values each
// load sat in its own basic block behind a wait: three quarters of the waves' cycles were spent parked on memory,
// profiles/r03_epilogue_pmc.txt).
template <int FAST>
__device__ __forceinline__ void sim_view_fast(SimView &S) {
    if (FAST) { S.dense = nullptr; S.nseg = 1; S.g16 = FAST == 1 ? 1u : 0u; }
}
template <int FAST>
__global__ __launch_bounds__(ST, 5) void pica2_kernel(SimBatch batch, const uint32_t *__restrict__ idx, uint32_t n_el,
                                                   const uint32_t *__restrict__ order, double threshold,
                                                   const uint64_t *__restrict__ seq_len, Pica2Out *__restrict__ out,
                                                   uint32_t *__restrict__ group_of, Pica2Split split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *rowsum = reinterpret_cast<double *>(lds_raw);
    uint32_t *grp = reinterpret_cast<uint32_t *>(rowsum + n_el);
    uint32_t *rep = grp + n_el;
    uint32_t *gsz = rep + n_el;
    __shared__ uint32_t sh_have;
    __shared__ uint64_t sh_npairs;
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);  // no identity memo here: with few groups it costs more than it saves
    sim_view_fast<FAST>(S);
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { sh_have = 0; sh_npairs = 0; }
    if (S.gram) {  // Gram problems: the diagonal a_i in LDS (behind the other arrays), else every identity costs three loads
        // 16-byte aligned (the launch's LDS size has 16 bytes of slack): join_blocks reads four diagonal entries at a time
        int32_t *diag_l = reinterpret_cast<int32_t *>((reinterpret_cast<uintptr_t>(gsz + n_el) + 15) & ~(uintptr_t)15);
        for (uint32_t i = tid; i < batch.n; i += ST) diag_l[i] = (int32_t)gram_at(S, i, i);
        S.diag = diag_l;
        __syncthreads();
    }
    // The `match` identity of a Gram problem is a function of the Hamming distance alone: memoise it (same arithmetic,
    // bit-identical values) instead of a division + decimal rounding per evaluation.  Large problems fill the memo up
    // front (the grouping itself makes up to n_el^2 / 2 evaluations), small ones only once they turn out to have many
    // groups (with a handful of groups filling it costs more than it saves)
    __shared__ double sim_tbl[SIM_TBL_N];
    const bool memo_first = n_el >= 1024 && !split.adj;
    if (memo_first) {
        sim_table_fill(S, sim_tbl, ST);
        __syncthreads();
    }
    // Step 1 (pica2.py:94-112)
    __shared__ __attribute__((aligned(8))) uint64_t rows_s[GG_ROWS];  // Step 1: candidate rows; Step 2: the groups' frequencies
    const uint32_t G = split.adj ? greedy_groups_bits<true>(split.adj + prob * (uint64_t)n_el * bit_words(n_el), S, idx, threshold, n_el,
                                                            grp, gsz, rep, reinterpret_cast<uint64_t *>(rowsum), rows_s)
                                 : greedy_groups(S, idx, n_el, threshold, order, grp, gsz, rep, reinterpret_cast<uint32_t *>(rowsum), rows_s);
    if (!memo_first && G >= 48) {  // many groups: the G(G-1)/2 representative pairs then cost a look-up each
        sim_table_fill(S, sim_tbl, ST);
        __syncthreads();
    }
    if (split.tab) {  // large problem: hand the groups over, Step 2 runs in pica2_rows_kernel
        uint32_t *tab = split_tab(split, prob, n_el);
        for (uint32_t g = tid; g < G; g += ST) { tab[g] = rep[g]; tab[n_el + g] = gsz[g]; }
        if (tid == 0) { tab[2 * n_el] = G; tab[2 * n_el + 1] = 0; tab[2 * n_el + 2] = 0; tab[2 * n_el + 3] = 0; }
        if (group_of)
            for (uint32_t i = tid; i < n_el; i += ST) group_of[prob * n_el + i] = grp[i];
        return;
    }
#if defined(IMPOP_PICA2_ABLATE) && (IMPOP_PICA2_ABLATE & 1)  // timing-only build: Step 1 alone
    if (tid == 0) { Pica2Out o; o.pi = 0; o.pi_site = 0; o.n_groups = G; o.pad = 0; o.sum_2pairs = 0; o.n_pairs = 0; out[prob] = o; }
    return;
#endif
    // Step 2-3 (pica2.py:118-154): sum over group pairs of 2*(1-sim(rep_i,rep_j))*f_i*f_j
    const double total = (double)n_el;
    uint32_t have = 0;
    uint64_t npairs = 0;
    if (G <= 64) {
        // few groups (the usual case): a thread per row, the row's pairs in the reference's order
        for (uint32_t i = tid; i < G; i += ST) {
            const uint32_t ri = idx ? idx[rep[i]] : rep[i];
            const double fi = (double)gsz[i] / total;
            double acc = 0.0;
            for (uint32_t j = i + 1; j < G; ++j) {
                const double s = sim_get(S, ri, idx ? idx[rep[j]] : rep[j]);
                if (s != s) continue;  // missing pair skipped (pica2.py:132-134)
                const double fj = (double)gsz[j] / total;
                const double pv = (1 - s) * fi * fj;
                acc += 2 * pv;
                have = 1;
                ++npairs;
            }
            rowsum[i] = acc;
        }
    } else {
        // many groups: a WAVE per row, lanes along the row — the representatives' Gram / table row is read coalesced
        // (a thread per row read one cache line per lane and pair: 4096 singleton groups cost 94 ms per window) —
        // lane partials in j order, fixed butterfly across lanes
        const uint32_t lane = tid & 63;
        // Gram problems of up to 512 elements, rows = positions: the row of a representative is walked over ALL positions
        // right of it with one 16-byte load per lane and 256 positions (as hfst_kernel does), against a per-POSITION
        // frequency table — f of the group a position represents, 0 for every other position, whose term then vanishes —
        // instead of one scalar Gram load per representative pair.  Every pair is present on a Gram problem, so the pair
        // count is G(G-1)/2.
        const bool quads = S.gram && S.diag && !idx && n_el <= GG_ROWS && gram_quads_aligned(S);
        if (quads) {
            double *fpos = reinterpret_cast<double *>(rows_s);
            for (uint32_t o = tid; o < GG_ROWS; o += ST) fpos[o] = 0.0;
            __syncthreads();
            for (uint32_t g = tid; g < G; g += ST) fpos[rep[g]] = (double)gsz[g] / total;
            __syncthreads();
            for (uint32_t i = wave_index(); i < G; i += ST / 64) {
                const uint32_t rr = rep[i];
                const double fi = (double)gsz[i] / total;
                const int64_t ar = S.diag[rr];
                double acc = 0.0;
                for (uint32_t o4 = 256 * (rr >> 8) + 4 * lane; o4 < n_el; o4 += 256) {
                    if (o4 + 3 <= rr) continue;  // entirely left of the diagonal
                    const uint64_t gp = (uint64_t)rr * S.ld + o4;  // o4 + 3 < ld: ld is a multiple of 4, o4 < n_el <= ld
                    i32q v = i32q{0, 0, 0, 0};  // nseg == 0: see join_blocks
                    for (uint32_t k = 0; k < S.nseg; ++k) v += gram_ld4(S, gp + k * S.seg_stride);
                    if (S.kind == IMPOP_IDENTITY_MATCH && S.tbl && S.W < (1ull << 30)) {
                        // `match` with the memo filled (it is, from 48 groups on): the same values as the general form below, in
                        // 32-bit arithmetic with the quad's four diagonal entries as one LDS read
                        i32q dg = i32q{0, 0, 0, 0};
                        if (o4 + 3 < n_el) dg = *reinterpret_cast<const i32q *>(S.diag + o4);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) dg[e] = o4 + e < n_el ? S.diag[o4 + e] : 0;
                        }
                        const int32_t ar32 = (int32_t)ar, add32 = (int32_t)S.add;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t o = o4 + e;
                            const bool live = o > rr && o < n_el;
                            const int32_t H = live ? ar32 + dg[e] - 2 * (v[e] + add32) : 0;
                            const double sv = (uint32_t)H < S.tbl_n ? S.tbl[H] : match_identity(S.W, (int64_t)H, S.round_digits);
                            const double fj = live ? fpos[o] : 0.0;
                            acc += 2 * ((1 - sv) * fi * fj);
                        }
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t o = o4 + e;
                        const bool live = o > rr && o < n_el;
                        // dead positions get the all-zero pair (a memo hit, never the division path) and frequency 0
                        const double sv = sim_from_gram(S, live ? (int64_t)v[e] + S.add : 0, live ? ar : 0, live ? (int64_t)S.diag[live ? o : rr] : 0);
                        const double fj = live ? fpos[o] : 0.0;
                        acc += 2 * ((1 - sv) * fi * fj);
                    }
                }
                acc = wave_sum_f64(acc);
                if (lane == 0) rowsum[i] = acc;
            }
            if (tid == 0 && G > 1) { have = 1; npairs = (uint64_t)G * (G - 1) / 2; }
        } else {
        // f_g = size / total once per group instead of a double-precision division per PAIR (the same quotient); the
        // table takes the candidate-row words of Step 1.  (Fewer instructions, no measurable change at 465 groups per
        // window: 4.9 ms per 4096 windows either way — the rows' Gram loads bound this loop, not its arithmetic.)
        double *ftab = G <= GG_ROWS ? reinterpret_cast<double *>(rows_s) : nullptr;
        if (ftab) {
            for (uint32_t g = tid; g < G; g += ST) ftab[g] = (double)gsz[g] / total;
            __syncthreads();
        }
        for (uint32_t i = wave_index(); i < G; i += ST / 64) {
            const uint32_t ri = idx ? idx[rep[i]] : rep[i];
            const double fi = (double)gsz[i] / total;
            double acc = 0.0;
            constexpr int P2_U = 8;  // identities in flight per lane: one at a time this loop is a chain of memory latencies
            for (uint32_t j0 = i + 1 + lane; j0 < G; j0 += 64 * P2_U) {
                double sv[P2_U];
#pragma unroll
                for (int u = 0; u < P2_U; ++u) {
                    const uint32_t j = j0 + 64 * u;
                    sv[u] = j < G ? sim_get(S, ri, idx ? idx[rep[j]] : rep[j]) : __builtin_nan("");
                }
#pragma unroll
                for (int u = 0; u < P2_U; ++u) {
                    const uint32_t j = j0 + 64 * u;
                    const double s = sv[u];
                    if (j >= G || s != s) continue;
                    const double fj = ftab ? ftab[j] : (double)gsz[j] / total;
                    acc += 2 * ((1 - s) * fi * fj);
                    have = 1;
                    ++npairs;
                }
            }
            acc = wave_sum_f64(acc);
            if (lane == 0) rowsum[i] = acc;
        }
        }
    }
    if (have) atomicOr(&sh_have, 1u);
    if (npairs) atomicAdd((unsigned long long *)&sh_npairs, (unsigned long long)npairs);
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        for (uint32_t i = 0; i < G; ++i) acc += rowsum[i];
        double pi = 0.0, pi_site = 0.0;
        if (n_el != 0 && sh_have) {
            pi = ((double)n_el / (double)(n_el - 1)) * acc;  // pica2.py:154
            const uint64_t L = seq_len ? seq_len[prob] : 0;
            pi_site = L ? pi / (double)L : __builtin_nan("");  // :163-164, None -> NaN
        }
        Pica2Out o;
        o.pi = pi; o.pi_site = pi_site; o.n_groups = G; o.pad = 0;
        o.sum_2pairs = acc; o.n_pairs = sh_npairs;
        out[prob] = o;
    }
    if (group_of)
        for (uint32_t i = tid; i < n_el; i += ST) group_of[prob * n_el + i] = grp[i];
}

// Step 2 of a large problem: workgroup (p, c) takes the group rows i = 4 c + wave, step 4 gridDim.y — a wave per row,
// lanes along the row, eight identities in flight per lane — and writes row sums; pair counts by integer atomics.
// dynamic LDS: diag[batch.n] int32 (Gram problems)
__global__ __launch_bounds__(ST) void pica2_rows_kernel(SimBatch batch, const uint32_t *__restrict__ idx, uint32_t n_el,
                                                        Pica2Split split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ double sim_tbl[SIM_TBL_N];
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (S.gram) {
        int32_t *diag_l = reinterpret_cast<int32_t *>(lds_raw);
        for (uint32_t i = tid; i < batch.n; i += ST) diag_l[i] = (int32_t)gram_at(S, i, i);
        S.diag = diag_l;
    }
    sim_table_fill(S, sim_tbl, ST);
    __syncthreads();
    uint32_t *tab = split_tab(split, prob, n_el);
    const uint32_t *rep = tab, *gsz = tab + n_el;
    const uint32_t G = tab[2 * n_el];
    double *rowsum = split.rowsum + prob * n_el;
    const double total = (double)n_el;
    uint32_t have = 0;
    uint64_t npairs = 0;
    constexpr int P2_U = 8;
    for (uint32_t i = wave_index() + (ST / 64) * blockIdx.y; i < G; i += (ST / 64) * gridDim.y) {
        const uint32_t ri = idx ? idx[rep[i]] : rep[i];
        const double fi = (double)gsz[i] / total;
        double acc = 0.0;
        for (uint32_t j0 = i + 1 + lane; j0 < G; j0 += 64 * P2_U) {
            double sv[P2_U];
#pragma unroll
            for (int u = 0; u < P2_U; ++u) {
                const uint32_t j = j0 + 64 * u;
                sv[u] = j < G ? sim_get(S, ri, idx ? idx[rep[j]] : rep[j]) : __builtin_nan("");
            }
#pragma unroll
            for (int u = 0; u < P2_U; ++u) {
                const uint32_t j = j0 + 64 * u;
                const double s = sv[u];
                if (j >= G || s != s) continue;  // missing pair skipped (pica2.py:132-134)
                const double fj = (double)gsz[j] / total;
                acc += 2 * ((1 - s) * fi * fj);
                have = 1;
                ++npairs;
            }
        }
        acc = wave_sum_f64(acc);
        if (lane == 0) rowsum[i] = acc;
    }
    npairs = wave_sum_u64(npairs);
    if (lane == 0 && npairs) {
        atomicOr(&tab[2 * n_el + 1], 1u);
        atomicAdd(reinterpret_cast<unsigned long long *>(tab + 2 * n_el + 2), (unsigned long long)npairs);
    }
    (void)have;
}

__global__ void pica2_finish_kernel(Pica2Split split, uint32_t n_el, uint64_t n_problems, const uint64_t *__restrict__ seq_len,
                                    Pica2Out *__restrict__ out) {
    const uint64_t prob = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (prob >= n_problems) return;
    const uint32_t *tab = split_tab(split, prob, n_el);
    const uint32_t G = tab[2 * n_el];
    const double *rowsum = split.rowsum + prob * n_el;
    double acc = 0.0;
    for (uint32_t i = 0; i + 1 < G; ++i) acc += rowsum[i];  // the last row has no pair to its right
    double pi = 0.0, pi_site = 0.0;
    if (n_el != 0 && tab[2 * n_el + 1]) {
        pi = ((double)n_el / (double)(n_el - 1)) * acc;  // pica2.py:154
        const uint64_t L = seq_len ? seq_len[prob] : 0;
        pi_site = L ? pi / (double)L : __builtin_nan("");
    }
    Pica2Out o;
    o.pi = pi; o.pi_site = pi_site; o.n_groups = G; o.pad = 0;
    o.sum_2pairs = acc;
    o.n_pairs = (uint64_t)tab[2 * n_el + 2] | ((uint64_t)tab[2 * n_el + 3] << 32);
    out[prob] = o;
}

// ---------------------------------------------------------------------------------------
// h-fst.calculate_diversity / calculate_fst (h-fst.py:130-249)
// One workgroup per problem; a WAVE takes a row r and its lanes run along the columns c > r, so the
// Gram row is read coalesced (the upper triangle is the part the Gram kernel writes); the pair (r, c)
// counts for pi_A, pi_B or Dxy according to the classes of its two ends.  Classes, the Gram diagonal
// and the identity memo live in LDS.  (Before: one thread per row walking all columns, every load its
// own cache line: 0.82 us per 465-haplotype window, a third of the all-pairs path.)
// Occupancy is what this kernel lives on: a wave walks its rows one after the other and every Gram row
// comes from HBM (written a whole launch ago), so the SIMDs need many resident waves to hide that latency.
// Measured on 4096 x 10 kb windows (tools/time_epilogue.py): with a 32 KB memo + 20 KB of class / diagonal
// arrays (3 workgroups per CU) the pair loop cost 0.30 us per window; with an 8 KB memo and arrays sized by
// n (8 workgroups per CU) 0.15 us; issuing the loads of 8-32 column chunks ahead of their use did not help
// (more registers, fewer waves).  Identities of pairs further apart than the memo reaches are computed
// directly.
typedef int i32v4 __attribute__((ext_vector_type(4)));
constexpr uint32_t HF_LDS_N = 4096;  // problems up to this many sequences keep class + Gram diagonal in LDS
// Few, large problems (4096 haplotypes x a handful of windows): `splits` workgroups share a problem's member rows
// (gridDim.y; rows k = 4 s + wave, step 4 splits) and leave partial sums in `part`; hfst_finish_kernel adds them up in
// split order and does the arithmetic below.  splits == 1: everything in this kernel, as for the usual many-windows case.
struct HfstPart {
    double a, b, x;
    uint64_t ca, cb, cx;
};
__global__ void hfst_finish_kernel(const HfstPart *__restrict__ part, uint32_t splits, uint64_t n_problems,
                                   const uint64_t *__restrict__ seq_len, HfstOut *__restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_problems) return;
    double a = 0.0, b = 0.0, x = 0.0;
    for (uint32_t s = 0; s < splits; ++s) { a += part[p * splits + s].a; b += part[p * splits + s].b; x += part[p * splits + s].x; }
    const HfstPart q = part[p * splits];
    hfst_outputs(a, b, x, q.ca, 0, q.cb, 0, q.cx, 0, seq_len ? seq_len[p] : 0, out + p);
}

template <int FAST>
__global__ __launch_bounds__(ST) void hfst_kernel(SimBatch batch, const uint8_t *__restrict__ in_a,
                                                  const uint8_t *__restrict__ in_b, const uint64_t *__restrict__ seq_len,
                                                  HfstOut *__restrict__ out, HfstPart *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_lds[];  // diag[n4] int32 | cls[n4] u8 | rows[n] u32 (n <= HF_LDS_N)
    __shared__ double shd[ST / 64];
    __shared__ uint64_t shu[ST / 64];
    __shared__ double sim_tbl[SIM_TBL_N];
    const uint64_t prob = blockIdx.x;
    SimView S = sim_view(batch, prob);
    sim_view_fast<FAST>(S);
    sim_table_fill(S, sim_tbl, ST);
    const uint32_t n = batch.n, tid = threadIdx.x;
    const bool cached = n <= HF_LDS_N;
    int32_t *diag_l = reinterpret_cast<int32_t *>(hf_lds);
    const uint32_t n4 = (n + 3) & ~3u;  // the Gram path reads classes / diagonal four at a time
    uint8_t *cls_l = hf_lds + (size_t)n4 * 4;
    auto cls_of = [&](uint32_t i) -> uint32_t {  // 1 = A only, 2 = B only, 0 = neither or both (h-fst.py:181-185)
        const bool a = in_a[i], b = in_b[i];
        return (a && !b) ? 1u : (b && !a) ? 2u : 0u;
    };
    if (cached) {
        for (uint32_t i = tid; i < n4; i += ST) {
            cls_l[i] = i < n ? (uint8_t)cls_of(i) : (uint8_t)0;
            if (S.gram) diag_l[i] = i < n ? (int32_t)gram_at(S, i, i) : 0;
        }
        if (S.gram) S.diag = diag_l;
    }
    __syncthreads();
    double accA = 0.0, accB = 0.0, accX = 0.0;
    uint64_t cA = 0, mA = 0, cB = 0, mB = 0, cX = 0, mX = 0;
    const uint32_t lane = tid & 63;
    auto tally = [&](uint32_t cr, uint32_t cc, double s) {
        const bool miss = s != s;
        const double d = 1 - s;
        if (cr != cc) { if (miss) ++mX; else { accX += d; ++cX; } }
        else if (cr == 1) { if (miss) ++mA; else { accA += d; ++cA; } }
        else { if (miss) ++mB; else { accB += d; ++cB; } }
    };
    if (S.gram && cached && gram_quads_aligned(S)) {
        // Gram problems (every pair present): member rows only, lanes own FIXED columns (4 per lane and 256-column
        // step, one 16-byte load per segment), the row's class is wave-uniform and the column classes sit in a packed
        // LDS word, so a pair costs a handful of branch-free instructions: Hamming distance -> memoised identity ->
        // masked adds into (same class, other class).  Pair counts are the class sizes' products — nothing to count.
        __shared__ uint32_t n_members, n_a, n_b;
        if (n <= 512) {
            // Up to 512 sequences (the window-statistics shape): a GATHER over the member columns.  The sweep below evaluates
            // every 4-column slot of every member row — 240 rows x 512 columns = 123 k slots per 465-haplotype window for the
            // 28.7 k pairs that count, ~60 instructions each, and that arithmetic (not memory: 0.62 ms per 4096 windows for
            // 1.8 GB) was the kernel's time.  Here the two classes' members are listed in ascending order once; a wave takes a
            // member row r and visits exactly the members right of it, class A then class B — which accumulator a visit feeds
            // is then uniform per loop, no class masks, no dead columns — with all of a row's loads issued before the first
            // is consumed.  Same pairs, same identities (sim_from_gram), another summation order (tolerance: INTEGRATION.md §4).
            __shared__ uint16_t colA[512], colB[512], prefA[512], prefB[512];
            if (tid < 64) {
                uint32_t na = 0, nb = 0;
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t i = i0 + tid;
                    const uint32_t cc = i < n ? cls_l[i] : 0u;
                    const uint64_t balA = __ballot(cc == 1), balB = __ballot(cc == 2), below = (1ull << tid) - 1ull;
                    const uint32_t pa = na + (uint32_t)__popcll(balA & below), pb = nb + (uint32_t)__popcll(balB & below);
                    if (cc == 1) colA[pa] = (uint16_t)i;
                    if (cc == 2) colB[pb] = (uint16_t)i;
                    if (i < n) {  // members of the class at positions <= i: where the members RIGHT of i start in the list
                        prefA[i] = (uint16_t)(pa + (cc == 1));
                        prefB[i] = (uint16_t)(pb + (cc == 2));
                    }
                    na += (uint32_t)__popcll(balA);
                    nb += (uint32_t)__popcll(balB);
                }
                if (tid == 0) { n_a = na; n_b = nb; n_members = na + nb; }
            }
            __syncthreads();
            const uint32_t na = n_a, nb = n_b, nmem = n_members;
            for (uint32_t k = wave_index(); k < nmem; k += ST / 64) {
                const uint32_t r = k < na ? colA[k] : colB[k - na];
                const uint32_t cr = k < na ? 1u : 2u;
                const int64_t ar = S.diag[r];
                const uint64_t g = (uint64_t)r * S.ld;
                double toA = 0.0, toB = 0.0;
                // the members right of r: list positions [s, cnt) of each class, 64 per step, four steps' loads in flight; the
                // step counts are wave-uniform (scalar branches), only the last step of a class has idle lanes
#pragma unroll
                for (int cls = 0; cls < 2; ++cls) {
                    const uint16_t *col = cls ? colB : colA;
                    const uint32_t s0 = cls ? prefB[r] : prefA[r], cnt = cls ? nb : na;
                    double to = 0.0;
                    for (uint32_t p0 = s0; p0 < cnt; p0 += 256) {
                        int32_t v[4];
                        uint32_t c[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            v[u] = 0; c[u] = 0xFFFFu;
                            if (p0 + 64u * u < cnt) {  // uniform
                                const uint32_t kk = p0 + 64u * u + lane;
                                if (kk < cnt) {
                                    c[u] = col[kk];
                                    for (uint32_t q = 0; q < S.nseg; ++q) v[u] += gram_ld1(S, g + q * S.seg_stride + c[u]);
                                }
                            }
                        }
                        if (S.kind == IMPOP_IDENTITY_MATCH && S.tbl && S.W < (1ull << 30)) {  // memoised `match`: 32-bit Hamming arithmetic
                            const int32_t ar32 = (int32_t)ar, add32 = (int32_t)S.add;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (p0 + 64u * u < cnt && c[u] != 0xFFFFu) {
                                    const int32_t H = ar32 + diag_l[c[u]] - 2 * (v[u] + add32);
                                    to += 1 - ((uint32_t)H < S.tbl_n ? S.tbl[H] : match_identity(S.W, (int64_t)H, S.round_digits));
                                }
                            continue;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (p0 + 64u * u < cnt && c[u] != 0xFFFFu)
                                to += 1 - sim_from_gram(S, (int64_t)v[u] + S.add, ar, (int64_t)diag_l[c[u]]);
                    }
                    if (cls) toB = to; else toA = to;
                }
                if (cr == 1) { accA += toA; accX += toB; }
                else { accB += toB; accX += toA; }
            }
            if (tid == 0) {
                const uint64_t a_ = na, b_ = nb;
                cA = a_ * (a_ - (a_ ? 1 : 0)) / 2; cB = b_ * (b_ - (b_ ? 1 : 0)) / 2; cX = a_ * b_;
            }
        } else {
        uint32_t *rows_l = reinterpret_cast<uint32_t *>(hf_lds + (((size_t)n4 * 5 + 15) & ~(size_t)15));  // member rows, ascending
        if (tid == 0) { n_members = 0; n_a = 0; n_b = 0; }
        __syncthreads();
        if (tid < 64) {  // one wave compacts the member list in order (ballot + prefix popcount)
            uint32_t base = 0, na = 0, nb = 0;
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + tid;
                const uint32_t cc = i < n ? cls_l[i] : 0u;
                const uint64_t bal = __ballot(cc != 0);
                if (cc) rows_l[base + (uint32_t)__popcll(bal & ((1ull << tid) - 1ull))] = i;
                base += (uint32_t)__popcll(bal);
                na += (uint32_t)__popcll(__ballot(cc == 1));
                nb += (uint32_t)__popcll(__ballot(cc == 2));
            }
            if (tid == 0) { n_members = base; n_a = na; n_b = nb; }
        }
        __syncthreads();
        const uint32_t nm = n_members;
        for (uint32_t k = wave_index() + (ST / 64) * blockIdx.y; k < nm; k += (ST / 64) * gridDim.y) {
            const uint32_t r = rows_l[k];
            const uint32_t cr = cls_l[r];
            const int64_t ar = S.diag[r];
            double same = 0.0, other = 0.0;
            for (uint32_t c4 = 4 * lane; c4 < n; c4 += 256) {
                if (c4 + 3 <= r) continue;  // entirely left of the diagonal
                const uint64_t g = (uint64_t)r * S.ld + c4;
                int64_t I[4] = {S.add, S.add, S.add, S.add};
                for (uint32_t q = 0; q < S.nseg; ++q) {
                    const gram_i32x4 v = gram_ld4(S, g + q * S.seg_stride);
                    I[0] += v.x; I[1] += v.y; I[2] += v.z; I[3] += v.w;
                }
                const uint32_t cls4 = *reinterpret_cast<const uint32_t *>(cls_l + c4);  // n <= ld, LDS padded to 4
                const i32v4 dg = *reinterpret_cast<const i32v4 *>(diag_l + c4);
                const int32_t dgv[4] = {dg.x, dg.y, dg.z, dg.w};
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e) {
                    const uint32_t c = c4 + e;
                    const uint32_t cc = (cls4 >> (8 * e)) & 0xFFu;
                    const bool live = (c > r) & (c < n) & (cc != 0);
                    // dead lanes (left of the diagonal, padding, non-members) get the all-zero pair: a memo hit, never
                    // the division + decimal-rounding path the whole wave would have to wait for
                    const double d = 1 - sim_from_gram(S, live ? I[e] : 0, live ? ar : 0, live ? (int64_t)dgv[e] : 0);
                    same += (live & (cc == cr)) ? d : 0.0;
                    other += (live & (cc != cr)) ? d : 0.0;
                }
            }
            if (cr == 1) accA += same; else accB += same;
            accX += other;
        }
        if (tid == 0) {
            const uint64_t na = n_a, nb = n_b;
            cA = na * (na - (na ? 1 : 0)) / 2; cB = nb * (nb - (nb ? 1 : 0)) / 2; cX = na * nb;
        }
        }
    } else
    for (uint32_t r = wave_index(); r < n; r += ST / 64) {
        const uint32_t cr = cached ? cls_l[r] : cls_of(r);  // wave-uniform
        if (!cr) continue;
        for (uint32_t c = r + 1 + lane; c < n; c += 64) {
            const uint32_t cc = cached ? cls_l[c] : cls_of(c);
            if (!cc) continue;
            tally(cr, cc, sim_get(S, r, c));
        }
    }
    accA = block_sum_f64(accA, shd); accB = block_sum_f64(accB, shd); accX = block_sum_f64(accX, shd);
    cA = block_sum_u64(cA, shu); mA = block_sum_u64(mA, shu);
    cB = block_sum_u64(cB, shu); mB = block_sum_u64(mB, shu);
    cX = block_sum_u64(cX, shu); mX = block_sum_u64(mX, shu);
    if (tid == 0) {
        if (gridDim.y > 1) {  // partial sums of this split; counts are the same in every split (analytic on Gram problems)
            HfstPart q;
            q.a = accA; q.b = accB; q.x = accX; q.ca = cA; q.cb = cB; q.cx = cX;
            part[prob * gridDim.y + blockIdx.y] = q;
        } else {
            hfst_outputs(accA, accB, accX, cA, mA, cB, mB, cX, mX, seq_len ? seq_len[prob] : 0, out + prob);
        }
    }
}

// ---------------------------------------------------------------------------------------
// scripts/hudson/hud.py "grouped" Fst (hud.py:64-128, 173-300): greedy groups INSIDE each
// population, frequency-weighted sums over group pairs with the first pair present in the table
// (members in sorted order) as the groups' similarity.
// hud.py:88-99 get_group_similarity: first (member of g1) x (member of g2) pair that is present
__device__ __forceinline__ double hud_first_found(const SimView &S, const uint32_t *ia, const uint32_t *ga, uint32_t ma, uint32_t g1,
                                         const uint32_t *ib, const uint32_t *gb, uint32_t mb, uint32_t g2) {
    for (uint32_t i = 0; i < ma; ++i) {
        if (ga[i] != g1) continue;
        for (uint32_t j = 0; j < mb; ++j) {
            if (gb[j] != g2) continue;
            const double v = sim_get(S, ia[i], ib[j]);
            if (v == v) return v;
        }
    }
    return __builtin_nan("");
}
// order_a / order_b (nullable): seed orders inside A / B as positions into ia / ib (see greedy_groups)
// dynamic LDS: rowsum[max(ma,mb)] f64 (grouping scratch first) | grpA[ma] | szA[ma] | repA[ma] | grpB[mb] | szB[mb] | repB[mb]
__global__ __launch_bounds__(ST, 5) void hud_grouped_kernel(SimBatch batch, const uint32_t *__restrict__ ia, uint32_t ma,
                                                         const uint32_t *__restrict__ ib, uint32_t mb,
                                                         const uint32_t *__restrict__ order_a,
                                                         const uint32_t *__restrict__ order_b, double threshold,
                                                         const uint64_t *__restrict__ seq_len, HfstOut *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t mx = ma > mb ? ma : mb;
    double *rowsum = reinterpret_cast<double *>(lds_raw);
    uint32_t *grpA = reinterpret_cast<uint32_t *>(rowsum + mx);
    uint32_t *szA = grpA + ma, *repA = szA + ma, *grpB = repA + ma, *szB = grpB + mb, *repB = szB + mb;
    __shared__ double sh_res[3];
    __shared__ uint64_t sh_miss[3];
    const uint64_t prob = blockIdx.x;
    const SimView S = sim_view(batch, prob);
    const uint32_t tid = threadIdx.x;
    __shared__ __attribute__((aligned(8))) uint64_t rows_s[GG_ROWS];
    const uint32_t GA = greedy_groups(S, ia, ma, threshold, order_a, grpA, szA, repA, reinterpret_cast<uint32_t *>(rowsum), rows_s);
    const uint32_t GB = greedy_groups(S, ib, mb, threshold, order_b, grpB, szB, repB, reinterpret_cast<uint32_t *>(rowsum), rows_s);
    if (tid < 3) sh_miss[tid] = 0;
    __syncthreads();
    // within A, within B (hud.py:101-128), then between (hud.py:235-263): one pass each
    for (int pass = 0; pass < 3; ++pass) {
        const uint32_t *r1 = pass == 1 ? repB : repA, *r2 = pass == 0 ? repA : repB;
        const uint32_t *i1 = pass == 1 ? ib : ia, *g1 = pass == 1 ? grpB : grpA, *s1 = pass == 1 ? szB : szA;
        const uint32_t *i2 = pass == 0 ? ia : ib, *g2 = pass == 0 ? grpA : grpB, *s2 = pass == 0 ? szA : szB;
        const uint32_t m1 = pass == 1 ? mb : ma, m2 = pass == 0 ? ma : mb;
        const uint32_t G1 = pass == 1 ? GB : GA, G2 = pass == 0 ? GA : GB;
        uint64_t miss = 0;
        for (uint32_t x = tid; x < G1; x += ST) {
            double acc = 0.0;
            for (uint32_t y = (pass == 2 ? 0 : x + 1); y < G2; ++y) {
                // the first pair the reference tries is (first member of g1, first member of g2): present on every Gram
                // problem and almost always in a .sim table; only if it is absent, search on in sorted order
                double sv = sim_get(S, i1[r1[x]], i2[r2[y]]);
                if (sv != sv) sv = hud_first_found(S, i1, g1, m1, x, i2, g2, m2, y);
                if (sv != sv) { ++miss; continue; }
                if (pass == 2) acc += (((double)s1[x] * (double)s2[y]) / ((double)m1 * (double)m2)) * (1 - sv);  // :255-256
                else acc += 2 * ((double)s1[x] / (double)m1) * ((double)s2[y] / (double)m2) * (1 - sv);        // :119-121
            }
            rowsum[x] = acc;
        }
        if (miss) atomicAdd((unsigned long long *)&sh_miss[pass], (unsigned long long)miss);
        __syncthreads();
        if (tid == 0) {
            double acc = 0.0;
            for (uint32_t x = 0; x < G1; ++x) acc += rowsum[x];
            if (pass < 2) acc = (m1 <= 1) ? 0.0 : acc * (double)m1 / (double)(m1 - 1);  // :106-107, :127
            sh_res[pass] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double pi_a = sh_res[0], pi_b = sh_res[1], dxy = sh_res[2];
        const double pi_xy = 0.5 * (pi_a + pi_b);
        const double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;
        const uint64_t L = seq_len ? seq_len[prob] : 0;
        HfstOut o;
        if (L > 0) {
            const double dl = (double)L;
            o.v[0] = fst; o.v[1] = pi_a / dl; o.v[2] = pi_b / dl; o.v[3] = pi_xy / dl; o.v[4] = dxy / dl; o.v[5] = (dxy - pi_xy) / dl;
        } else {
            o.v[0] = fst; o.v[1] = pi_a; o.v[2] = pi_b; o.v[3] = pi_xy; o.v[4] = dxy; o.v[5] = dxy - pi_xy;
        }
        o.cnt[0] = GA; o.cnt[1] = sh_miss[0]; o.cnt[2] = GB; o.cnt[3] = sh_miss[1];
        o.cnt[4] = (uint64_t)GA * GB - sh_miss[2]; o.cnt[5] = sh_miss[2];
        out[prob] = o;
    }
}

// ---------------------------------------------------------------------------------------
// af.cluster (af.py:35-44): adjacency bits, then min-label propagation with pointer jumping
__global__ void af_adjacency_kernel(SimBatch batch, double threshold, uint32_t words, uint32_t *__restrict__ adj) {
    const uint32_t n = batch.n;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * words) return;
    const uint32_t i = (uint32_t)(t / words), w = (uint32_t)(t % words);
    const SimView S = sim_view(batch, 0);
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint32_t j = 32 * w + b;
        if (j >= n) break;
        const double v = sim_get(S, i, j);
        if (v == v && v >= threshold) bits |= 1u << b;  // non-strict (af.py:38)
    }
    adj[t] = bits;
}

// dynamic LDS: label[n] | size[n] | rank[n]
__global__ __launch_bounds__(ST) void af_components_kernel(uint32_t n, uint32_t words, const uint32_t *__restrict__ adj,
                                                           uint32_t *__restrict__ cluster_of, uint32_t *__restrict__ sizes,
                                                           uint32_t *__restrict__ n_clusters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *label = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *size = label + n;
    uint32_t *rank = size + n;
    __shared__ uint32_t changed, K;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += ST) { label[i] = i; size[i] = 0; }
    __syncthreads();
    for (uint32_t iter = 0; iter <= n; ++iter) {  // bounded: labels only decrease
        if (tid == 0) changed = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += ST) {
            uint32_t m = label[i];
            for (uint32_t w = 0; w < words; ++w) {
                uint32_t bits = adj[(uint64_t)i * words + w];
                while (bits) {
                    const uint32_t j = 32 * w + (uint32_t)(__ffs(bits) - 1);
                    bits &= bits - 1;
                    const uint32_t lj = label[j];
                    if (lj < m) m = lj;
                    // symmetric relation: pull i's label into j as well (rows may be one-sided
                    // when only one orientation of a pair is present in the table)
                    if (label[i] < lj) { atomicMin(&label[j], label[i]); changed = 1; }
                }
            }
            if (m < label[i]) { atomicMin(&label[i], m); changed = 1; }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += ST) {
            const uint32_t l = label[i], ll = label[l];
            if (ll < l) { atomicMin(&label[i], ll); changed = 1; }
        }
        __syncthreads();
        const uint32_t c = changed;
        __syncthreads();
        if (!c) break;
    }
    for (uint32_t i = tid; i < n; i += ST) atomicAdd(&size[label[i]], 1u);
    if (tid == 0) K = 0;
    __syncthreads();
    // order roots by (-size, smallest member) (af.py:43); the root label IS the smallest member
    for (uint32_t r = tid; r < n; r += ST) {
        if (!size[r]) continue;
        uint32_t rk = 0;
        for (uint32_t q = 0; q < n; ++q) {
            if (!size[q] || q == r) continue;
            if (size[q] > size[r] || (size[q] == size[r] && q < r)) ++rk;
        }
        rank[r] = rk;
        sizes[rk] = size[r];
        atomicAdd(&K, 1u);
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += ST) cluster_of[i] = rank[label[i]];
    if (tid == 0) *n_clusters = K;
}

// ---------------------------------------------------------------------------------------
__global__ void tajima_kernel(const int64_t *__restrict__ n, const double *__restrict__ S, const double *__restrict__ pi,
                              uint64_t count, double *__restrict__ D, double *__restrict__ comps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const TajConsts c = tajima_consts(n[i]);
    double num, den;
    D[i] = tajima_d_from(c, S[i], pi[i], &num, &den);
    if (comps) {
        double *o = comps + i * 10;
        o[0] = c.a1; o[1] = c.a2; o[2] = c.b1; o[3] = c.b2; o[4] = c.c1; o[5] = c.c2; o[6] = c.e1; o[7] = c.e2;
        o[8] = num; o[9] = den;
    }
}

__global__ void py_round_kernel(const double *__restrict__ x, uint64_t count, int nd, double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = py_round(x[i], nd);
}

// dynamic LDS a kernel may ask for: the 160 KB of a CDNA4 workgroup minus what its static arrays take (and 1 KB of slack)
static size_t dynamic_lds_room(const void *kernel) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return 128 * 1024;
    const size_t total = 160 * 1024, used = fa.sharedSizeBytes + 1024;
    return used < total ? total - used : 0;
}

// which kernel variant a batch may take (pica2_kernel / hfst_kernel FAST): Gram problems, one matrix per problem at index p
static int sim_batch_fast(const SimBatch &b) {
    static const bool off = [] { const char *e = getenv("IMPOP_EPILOGUE_FAST"); return e && e[0] == '0'; }();  // A/B and test switch
    if (off || !b.gram || b.dense || b.seg_first || b.seg_count) return 0;
    return b.g16 ? 1 : 2;
}

int launch_pica2(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_idx, uint32_t n_el,
                 const uint32_t *d_order, double threshold, const uint64_t *d_seq_len, Pica2Out *d_out,
                 uint32_t *d_group_of) {
    if (!n_problems) return IMPOP_OK;
    REQUIRE(n_problems < 0x7FFFFFFFull, "pica2: too many problems");
    if (pica2_small_applies(b, n_el, d_order, d_group_of))  // the window-statistics shape: stats_small.hip
        return launch_pica2_small(ctx, b, n_problems, d_idx, n_el, threshold, d_seq_len, d_out);
    const size_t lds = (size_t)n_el * (8 + 12) + (b.gram ? (size_t)b.n * 4 : 0) + 16;
    // 160 KB per workgroup, minus the kernel's static arrays (identity memo, candidate rows of the grouping, counters:
    // ~14 KB) — asked of the runtime, so that the limit follows the kernel
    static const size_t lds_room = dynamic_lds_room((const void *)pica2_kernel<0>);  // the variants share their static arrays
    REQUIRE(lds <= lds_room, "pica2: %u elements exceed the LDS-resident grouping limit (about 7300; 6100 on Gram problems)", n_el);
    REQUIRE(n_problems < 0x7FFFFFFFull, "pica2: too many problems");
    // few large problems: Step 2 is split over row chunks (about 8 workgroups per CU in total, >= 16 rows each)
    uint32_t chunks = 1;
    if (n_el >= 1024) {
        const uint64_t want = 8ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
        chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want / n_problems, 1), n_el / 16);
        if (chunks > 1024) chunks = 1024;
    }
    Pica2Split split{nullptr, nullptr, nullptr};
    if (chunks > 1) {
        const size_t tab_bytes = ((size_t)n_problems * (2ull * n_el + 4) * 4 + 255) / 256 * 256;
        const size_t sum_bytes = ((size_t)n_problems * n_el * 8 + 255) / 256 * 256;
        // seeds in position order (no set order handed in): the join tests of Step 1 go to pica2_adj_kernel, one bit each
        const size_t adj_bytes = (size_t)n_problems * n_el * bit_words(n_el) * 8;
        const bool bits = !d_order && n_el <= 8192 && adj_bytes <= ((size_t)2 << 30);
        void *aux = nullptr;
        const int arc = ctx_aux(ctx, 1, tab_bytes + sum_bytes + (bits ? adj_bytes : 0), &aux);
        if (arc) return arc;
        split.tab = reinterpret_cast<uint32_t *>(aux);
        split.rowsum = reinterpret_cast<double *>((char *)aux + tab_bytes);
        if (bits) {
            split.adj = reinterpret_cast<uint64_t *>((char *)aux + tab_bytes + sum_bytes);
            hipLaunchKernelGGL(pica2_adj_kernel, dim3((uint32_t)n_problems, chunks), dim3(ST), b.gram ? (size_t)b.n * 4 : 0, ctx->stream,
                               b, d_idx, n_el, threshold, split);
            HIP_TRY(hipGetLastError());
        }
    }
    SimBatch be = b;
    be.err = ctx->d_err;  // the grouping's progress bound reports here (ctx_err_fetch / ctx_err_result in the caller)
    const int fast = sim_batch_fast(b);
    const void *kfn = fast == 1 ? (const void *)pica2_kernel<1> : fast == 2 ? (const void *)pica2_kernel<2> : (const void *)pica2_kernel<0>;
    if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (fast == 1)
        hipLaunchKernelGGL(pica2_kernel<1>, dim3((uint32_t)n_problems), dim3(ST), lds, ctx->stream, be, d_idx, n_el, d_order, threshold,
                           d_seq_len, d_out, d_group_of, split);
    else if (fast == 2)
        hipLaunchKernelGGL(pica2_kernel<2>, dim3((uint32_t)n_problems), dim3(ST), lds, ctx->stream, be, d_idx, n_el, d_order, threshold,
                           d_seq_len, d_out, d_group_of, split);
    else
        hipLaunchKernelGGL(pica2_kernel<0>, dim3((uint32_t)n_problems), dim3(ST), lds, ctx->stream, be, d_idx, n_el, d_order, threshold,
                           d_seq_len, d_out, d_group_of, split);
    HIP_TRY(hipGetLastError());
    if (chunks > 1) {
        hipLaunchKernelGGL(pica2_rows_kernel, dim3((uint32_t)n_problems, chunks), dim3(ST), b.gram ? (size_t)b.n * 4 : 0, ctx->stream, b,
                           d_idx, n_el, split);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(pica2_finish_kernel, dim3((uint32_t)((n_problems + 63) / 64)), dim3(64), 0, ctx->stream, split, n_el,
                           n_problems, d_seq_len, d_out);
        HIP_TRY(hipGetLastError());
    }
    return IMPOP_OK;
}

int launch_hfst(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint8_t *d_in_a, const uint8_t *d_in_b,
                const uint64_t *d_seq_len, HfstOut *d_out) {
    if (!n_problems) return IMPOP_OK;
    REQUIRE(n_problems < 0x7FFFFFFFull, "hfst: too many problems");
    if (hfst_small_applies(b)) return launch_hfst_small(ctx, b, n_problems, d_in_a, d_in_b, d_seq_len, d_out);  // stats_small.hip
    // diag[n4] int32 | cls[n4] u8 | (16-aligned) member rows[n] u32, n4 = n rounded up to 4
    const size_t n4 = ((size_t)b.n + 3) & ~(size_t)3;
    const size_t lds = b.n <= HF_LDS_N ? ((n4 * 5 + 15) & ~(size_t)15) + (size_t)b.n * 4 + 16 : 16;
    // one workgroup per problem fills the chip when there are thousands of problems; a few LARGE Gram problems are split
    // over several workgroups each (about 8 per CU in total, never more than one per 16 rows)
    uint32_t splits = 1;
    if (b.gram && b.n >= 1024 && b.n <= HF_LDS_N && (b.ld & 3u) == 0) {
        const uint64_t want = 8ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
        splits = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want / n_problems, 1), b.n / 16);
        if (splits > 1024) splits = 1024;
    }
    HfstPart *d_part = nullptr;
    if (splits > 1) {
        void *aux = nullptr;
        const int arc = ctx_aux(ctx, 0, (size_t)n_problems * splits * sizeof(HfstPart), &aux);
        if (arc) return arc;
        d_part = reinterpret_cast<HfstPart *>(aux);
    }
    const int fast = sim_batch_fast(b);
    if (fast == 1)
        hipLaunchKernelGGL(hfst_kernel<1>, dim3((uint32_t)n_problems, splits), dim3(ST), lds, ctx->stream, b, d_in_a, d_in_b, d_seq_len,
                           d_out, d_part);
    else if (fast == 2)
        hipLaunchKernelGGL(hfst_kernel<2>, dim3((uint32_t)n_problems, splits), dim3(ST), lds, ctx->stream, b, d_in_a, d_in_b, d_seq_len,
                           d_out, d_part);
    else
        hipLaunchKernelGGL(hfst_kernel<0>, dim3((uint32_t)n_problems, splits), dim3(ST), lds, ctx->stream, b, d_in_a, d_in_b, d_seq_len,
                           d_out, d_part);
    HIP_TRY(hipGetLastError());
    if (splits > 1) {
        hipLaunchKernelGGL(hfst_finish_kernel, dim3((uint32_t)((n_problems + 63) / 64)), dim3(64), 0, ctx->stream, d_part, splits,
                           n_problems, d_seq_len, d_out);
        HIP_TRY(hipGetLastError());
    }
    return IMPOP_OK;
}

int launch_hud_grouped(impop_ctx *ctx, const SimBatch &b, uint64_t n_problems, const uint32_t *d_ia, uint32_t ma,
                       const uint32_t *d_ib, uint32_t mb, const uint32_t *d_order_a, const uint32_t *d_order_b,
                       double threshold, const uint64_t *d_seq_len, HfstOut *d_out) {
    if (!n_problems) return IMPOP_OK;
    REQUIRE(n_problems < 0x7FFFFFFFull, "grouped Fst: too many problems");
    const size_t lds = (size_t)std::max(ma, mb) * 8 + ((size_t)ma + mb) * 12 + 16;
    static const size_t lds_room = dynamic_lds_room((const void *)hud_grouped_kernel);
    REQUIRE(lds <= lds_room, "grouped Fst: populations too large for the LDS-resident grouping");
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)hud_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SimBatch be = b;
    be.err = ctx->d_err;
    hipLaunchKernelGGL(hud_grouped_kernel, dim3((uint32_t)n_problems), dim3(ST), lds, ctx->stream, be, d_ia, ma, d_ib, mb,
                       d_order_a, d_order_b, threshold, d_seq_len, d_out);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

int launch_af(impop_ctx *ctx, const SimBatch &b, double threshold, uint32_t *d_adj, uint32_t *d_cluster_of,
              uint32_t *d_sizes, uint32_t *d_nclusters) {
    const uint32_t n = b.n, words = (n + 31) / 32;
    const size_t lds = (size_t)n * 12 + 16;
    REQUIRE(lds <= 150 * 1024, "af: %u samples exceed the LDS-resident clustering limit (12700)", n);
    if (n) {
        const uint64_t total = (uint64_t)n * words;
        hipLaunchKernelGGL(af_adjacency_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, b,
                           threshold, words, d_adj);
        HIP_TRY(hipGetLastError());
    }
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)af_components_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(af_components_kernel, dim3(1), dim3(ST), lds, ctx->stream, n, words, d_adj, d_cluster_of, d_sizes,
                       d_nclusters);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

// scratch carve helper
struct Carve {
    char *base;
    size_t off = 0;
    explicit Carve(void *p) : base((char *)p) {}
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) / 256 * 256;
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};
static size_t carve_size(std::initializer_list<size_t> sizes) {
    size_t t = 0;
    for (size_t s : sizes) t = (t + 255) / 256 * 256 + s;
    return t + 256;
}

}  // namespace impop

using namespace impop;

// seed_rank -> order (inverse permutation) restricted to `members` (positions 0..m of the member list);
// ranks only need to be distinct among the members
static int seed_order_of(const uint32_t *seed_rank, const std::vector<uint32_t> &members, std::vector<uint32_t> &order,
                         const char *fn) {
    const uint32_t m = (uint32_t)members.size();
    order.resize(m);
    for (uint32_t k = 0; k < m; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t x, uint32_t y) { return seed_rank[members[x]] < seed_rank[members[y]]; });
    for (uint32_t k = 1; k < m; ++k)
        REQUIRE(seed_rank[members[order[k - 1]]] != seed_rank[members[order[k]]],
                "%s: seed_rank must be distinct among the elements it orders (rank %u occurs twice)", fn,
                seed_rank[members[order[k]]]);
    return IMPOP_OK;
}

IMPOP_API int impop_pi_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, double threshold, int round_digits,
                                     uint64_t seq_len, const uint32_t *seed_rank, double *pi, double *pi_site,
                                     uint32_t *group_of, uint32_t *n_groups, impop_pica2_detail *detail) {
    REQUIRE(ctx, "impop_pi_from_identity: ctx is NULL");
    REQUIRE(n == 0 || ident, "impop_pi_from_identity: ident is NULL");
    REQUIRE(round_digits <= 19, "impop_pi_from_identity: round_digits > 19 unsupported");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint32_t> order;
    if (seed_rank && n) {
        std::vector<uint32_t> all(n);
        for (uint32_t i = 0; i < n; ++i) all[i] = i;
        int rc0 = seed_order_of(seed_rank, all, order, "impop_pi_from_identity");
        if (rc0) return rc0;
    }
    const size_t nn = (size_t)n * n;
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({nn * 8, 8, sizeof(Pica2Out), (size_t)n * 4, (size_t)n * 4}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_id = cv.take<double>(nn ? nn : 1);
    uint64_t *d_L = cv.take<uint64_t>(1);
    Pica2Out *d_out = cv.take<Pica2Out>(1);
    uint32_t *d_grp = cv.take<uint32_t>(n ? n : 1);
    uint32_t *d_order = cv.take<uint32_t>(n ? n : 1);
    if (nn) HIP_TRY(hipMemcpyAsync(d_id, ident, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_L, &seq_len, 8, hipMemcpyHostToDevice, ctx->stream));
    if (!order.empty()) HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    SimBatch b{};
    b.dense = d_id; b.gram = nullptr; b.stride = nn; b.ld = n; b.n = n; b.W = nullptr; b.kind = 0;
    b.round_digits = round_digits < 0 ? -1 : round_digits;
    rc = launch_pica2(ctx, b, 1, nullptr, n, order.empty() ? nullptr : d_order, threshold, d_L, d_out, d_grp);
    if (rc) return rc;
    Pica2Out o;
    std::vector<uint32_t> g(n ? n : 1);
    HIP_TRY(hipMemcpyAsync(&o, d_out, sizeof o, hipMemcpyDeviceToHost, ctx->stream));
    if (n) HIP_TRY(hipMemcpyAsync(g.data(), d_grp, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    rc = ctx_err_fetch(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // `order` (pageable) must outlive its copy
    rc = ctx_err_result(ctx, "impop_pi_from_identity");
    if (rc) return rc;
    if (pi) *pi = o.pi;
    if (pi_site) *pi_site = o.pi_site;
    if (n_groups) *n_groups = o.n_groups;
    if (group_of && n) memcpy(group_of, g.data(), (size_t)n * 4);
    if (detail) { detail->sum_2pairs = o.sum_2pairs; detail->n_pairs_with_data = o.n_pairs; }
    return IMPOP_OK;
}

// the "Step 2" table of pica2's log (pica2.py:125-145): for groups i < j, identity of the two representatives
// (rounded like the analysis; NaN = pair absent, the line the reference replaces by a warning) and the term
// (1 - sim) * f_i * f_j.  One thread per pair.
__global__ void pica2_pairs_kernel(SimBatch batch, const uint32_t *__restrict__ rep, const uint32_t *__restrict__ gsz,
                                   uint32_t G, uint32_t total, double *__restrict__ sims, double *__restrict__ values) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t np = (uint64_t)G * (G - 1) / 2;
    if (t >= np) return;
    // row i of the strict upper triangle: the largest i with i(2G-i-1)/2 <= t
    uint32_t i = (uint32_t)(((2.0 * G - 1) - sqrt((2.0 * G - 1) * (2.0 * G - 1) - 8.0 * (double)t)) / 2.0);
    while (i > 0 && (uint64_t)i * (2ull * G - i - 1) / 2 > t) --i;
    while ((uint64_t)(i + 1) * (2ull * G - i - 2) / 2 <= t) ++i;
    const uint32_t j = (uint32_t)(t - (uint64_t)i * (2ull * G - i - 1) / 2) + i + 1;
    const SimView S = sim_view(batch, 0);
    const double sv = sim_get(S, rep[i], rep[j]);
    sims[t] = sv;
    const double fi = (double)gsz[i] / (double)total, fj = (double)gsz[j] / (double)total;  // pica2.py:137-138
    values[t] = (1 - sv) * fi * fj;                                                       // :139
}

IMPOP_API int impop_pica2_pair_terms(impop_ctx *ctx, const double *ident, uint32_t n, int round_digits,
                                     const uint32_t *rep, const uint32_t *group_size, uint32_t n_groups,
                                     double *sims_out, double *values_out) {
    REQUIRE(ctx, "impop_pica2_pair_terms: ctx is NULL");
    REQUIRE(round_digits <= 19, "impop_pica2_pair_terms: round_digits > 19 unsupported");
    if (n_groups < 2) return IMPOP_OK;
    REQUIRE(ident && rep && group_size && sims_out && values_out, "impop_pica2_pair_terms: NULL argument");
    uint64_t total = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        REQUIRE(rep[g] < n, "impop_pica2_pair_terms: representative %u out of range", rep[g]);
        total += group_size[g];
    }
    REQUIRE(total > 0 && total <= 0xFFFFFFFFull, "impop_pica2_pair_terms: bad group sizes");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const uint64_t np = (uint64_t)n_groups * (n_groups - 1) / 2;
    REQUIRE((np + 255) / 256 < 0x7FFFFFFFull, "impop_pica2_pair_terms: too many group pairs");
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({nn * 8, (size_t)n_groups * 4, (size_t)n_groups * 4, np * 8, np * 8}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_id = cv.take<double>(nn);
    uint32_t *d_rep = cv.take<uint32_t>(n_groups), *d_sz = cv.take<uint32_t>(n_groups);
    double *d_s = cv.take<double>(np), *d_v = cv.take<double>(np);
    HIP_TRY(hipMemcpyAsync(d_id, ident, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_rep, rep, (size_t)n_groups * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_sz, group_size, (size_t)n_groups * 4, hipMemcpyHostToDevice, ctx->stream));
    SimBatch b{};
    b.dense = d_id; b.stride = nn; b.ld = n; b.n = n; b.round_digits = round_digits < 0 ? -1 : round_digits;
    hipLaunchKernelGGL(pica2_pairs_kernel, dim3((uint32_t)((np + 255) / 256)), dim3(256), 0, ctx->stream, b, d_rep, d_sz,
                       n_groups, (uint32_t)total, d_s, d_v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sims_out, d_s, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(values_out, d_v, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_fst_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, const uint8_t *in_a,
                                      const uint8_t *in_b, uint64_t seq_len, int round_digits, double *out,
                                      uint64_t *counts) {
    REQUIRE(ctx && out, "impop_fst_from_identity: NULL argument");
    REQUIRE(n == 0 || (ident && in_a && in_b), "impop_fst_from_identity: NULL input");
    REQUIRE(round_digits <= 19, "impop_fst_from_identity: round_digits > 19 unsupported");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({nn * 8, 8, sizeof(HfstOut), (size_t)n, (size_t)n}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_id = cv.take<double>(nn ? nn : 1);
    uint64_t *d_L = cv.take<uint64_t>(1);
    HfstOut *d_out = cv.take<HfstOut>(1);
    uint8_t *d_a = cv.take<uint8_t>(n ? n : 1);
    uint8_t *d_b = cv.take<uint8_t>(n ? n : 1);
    if (nn) HIP_TRY(hipMemcpyAsync(d_id, ident, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n) {
        HIP_TRY(hipMemcpyAsync(d_a, in_a, n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(d_b, in_b, n, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(d_L, &seq_len, 8, hipMemcpyHostToDevice, ctx->stream));
    SimBatch b{};
    b.dense = d_id; b.stride = nn; b.ld = n; b.n = n; b.round_digits = round_digits < 0 ? -1 : round_digits;
    rc = launch_hfst(ctx, b, 1, d_a, d_b, d_L, d_out);
    if (rc) return rc;
    HfstOut o;
    HIP_TRY(hipMemcpyAsync(&o, d_out, sizeof o, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 6; ++k) out[k] = o.v[k];
    if (counts)
        for (int k = 0; k < 6; ++k) counts[k] = o.cnt[k];
    return IMPOP_OK;
}

IMPOP_API int impop_cluster_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, double threshold,
                                          uint32_t *cluster_of, uint32_t *n_clusters, uint32_t *sizes) {
    REQUIRE(ctx, "impop_cluster_from_identity: ctx is NULL");
    REQUIRE(n == 0 || (ident && cluster_of), "impop_cluster_from_identity: NULL argument");
    if (n == 0) {
        if (n_clusters) *n_clusters = 0;
        return IMPOP_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const uint32_t words = (n + 31) / 32;
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({nn * 8, (size_t)n * words * 4, (size_t)n * 4, (size_t)n * 4, 4}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_id = cv.take<double>(nn);
    uint32_t *d_adj = cv.take<uint32_t>((size_t)n * words);
    uint32_t *d_cl = cv.take<uint32_t>(n);
    uint32_t *d_sz = cv.take<uint32_t>(n);
    uint32_t *d_k = cv.take<uint32_t>(1);
    HIP_TRY(hipMemcpyAsync(d_id, ident, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_sz, 0, (size_t)n * 4, ctx->stream));
    SimBatch b{};
    b.dense = d_id; b.stride = nn; b.ld = n; b.n = n; b.round_digits = -1;
    rc = launch_af(ctx, b, threshold, d_adj, d_cl, d_sz, d_k);
    if (rc) return rc;
    std::vector<uint32_t> sz(n);
    uint32_t K = 0;
    HIP_TRY(hipMemcpyAsync(cluster_of, d_cl, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(sz.data(), d_sz, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&K, d_k, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_clusters) *n_clusters = K;
    if (sizes) memcpy(sizes, sz.data(), (size_t)n * 4);
    return IMPOP_OK;
}

IMPOP_API int impop_tajimas_d(impop_ctx *ctx, const int64_t *n, const double *S, const double *pi, uint64_t count,
                              double *D, double *comps) {
    REQUIRE(ctx, "impop_tajimas_d: ctx is NULL");
    if (!count) return IMPOP_OK;
    REQUIRE(n && S && pi && D, "impop_tajimas_d: NULL argument");
    for (uint64_t i = 0; i < count; ++i) {
        if (n[i] < 2) { set_error("n must be >= 2"); return IMPOP_E_INVALID; }                    // tj_d.py:48-49
        if (S[i] < 0 || pi[i] < 0) { set_error("S and pi must be non-negative"); return IMPOP_E_INVALID; }  // :50-51
    }
    HIP_TRY(hipSetDevice(ctx->device));
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({count * 8, count * 8, count * 8, count * 8, count * 80}), &d);
    if (rc) return rc;
    Carve cv(d);
    int64_t *d_n = cv.take<int64_t>(count);
    double *d_S = cv.take<double>(count), *d_pi = cv.take<double>(count), *d_D = cv.take<double>(count);
    double *d_c = cv.take<double>(count * 10);
    HIP_TRY(hipMemcpyAsync(d_n, n, count * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_S, S, count * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_pi, pi, count * 8, hipMemcpyHostToDevice, ctx->stream));
    REQUIRE((count + 63) / 64 < 0x7FFFFFFFull, "impop_tajimas_d: too many triples");
    hipLaunchKernelGGL(tajima_kernel, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, ctx->stream, d_n, d_S, d_pi, count,
                       d_D, comps ? d_c : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(D, d_D, count * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (comps) HIP_TRY(hipMemcpyAsync(comps, d_c, count * 80, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_py_round(impop_ctx *ctx, const double *x, uint64_t count, int ndigits, double *out) {
    REQUIRE(ctx, "impop_py_round: ctx is NULL");
    if (!count) return IMPOP_OK;
    REQUIRE(x && out, "impop_py_round: NULL argument");
    REQUIRE(ndigits >= 0 && ndigits <= 19, "impop_py_round: ndigits must be 0..19");
    HIP_TRY(hipSetDevice(ctx->device));
    void *d = nullptr;
    int rc = ctx_scratch(ctx, carve_size({count * 8, count * 8}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_x = cv.take<double>(count), *d_o = cv.take<double>(count);
    HIP_TRY(hipMemcpyAsync(d_x, x, count * 8, hipMemcpyHostToDevice, ctx->stream));
    REQUIRE((count + 255) / 256 < 0x7FFFFFFFull, "impop_py_round: too many values");
    hipLaunchKernelGGL(py_round_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_x, count,
                       ndigits, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_o, count * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return IMPOP_OK;
}

IMPOP_API int impop_fst_grouped_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, const uint8_t *in_a,
                                              const uint8_t *in_b, double threshold, uint64_t seq_len, int round_digits,
                                              const uint32_t *seed_rank, double *out, uint64_t *counts) {
    REQUIRE(ctx && out, "impop_fst_grouped_from_identity: NULL argument");
    REQUIRE(n == 0 || (ident && in_a && in_b), "impop_fst_grouped_from_identity: NULL input");
    REQUIRE(round_digits <= 19, "impop_fst_grouped_from_identity: round_digits > 19 unsupported");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint32_t> ia, ib;
    for (uint32_t i = 0; i < n; ++i) {  // members of both populations leave both (hud.py:186-190)
        const bool ov = in_a[i] && in_b[i];
        if (in_a[i] && !ov) ia.push_back(i);
        if (in_b[i] && !ov) ib.push_back(i);
    }
    const uint32_t ma = (uint32_t)ia.size(), mb = (uint32_t)ib.size();
    const size_t lds = (size_t)std::max(ma, mb) * 8 + ((size_t)ma + mb) * 12 + 16;
    REQUIRE(lds <= 150 * 1024, "grouped Fst: populations too large for the LDS-resident grouping");
    const size_t nn = (size_t)n * n;
    void *d = nullptr;
    std::vector<uint32_t> oa, ob;
    if (seed_rank) {
        int rc0 = seed_order_of(seed_rank, ia, oa, "impop_fst_grouped_from_identity");
        if (!rc0) rc0 = seed_order_of(seed_rank, ib, ob, "impop_fst_grouped_from_identity");
        if (rc0) return rc0;
    }
    int rc = ctx_scratch(ctx, carve_size({nn * 8, 8, sizeof(HfstOut), (size_t)ma * 4, (size_t)mb * 4, (size_t)ma * 4,
                                          (size_t)mb * 4}), &d);
    if (rc) return rc;
    Carve cv(d);
    double *d_id = cv.take<double>(nn ? nn : 1);
    uint64_t *d_L = cv.take<uint64_t>(1);
    HfstOut *d_out = cv.take<HfstOut>(1);
    uint32_t *d_ia = cv.take<uint32_t>(ma ? ma : 1), *d_ib = cv.take<uint32_t>(mb ? mb : 1);
    uint32_t *d_oa = cv.take<uint32_t>(ma ? ma : 1), *d_ob = cv.take<uint32_t>(mb ? mb : 1);
    if (!oa.empty()) HIP_TRY(hipMemcpyAsync(d_oa, oa.data(), (size_t)ma * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!ob.empty()) HIP_TRY(hipMemcpyAsync(d_ob, ob.data(), (size_t)mb * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nn) HIP_TRY(hipMemcpyAsync(d_id, ident, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    if (ma) HIP_TRY(hipMemcpyAsync(d_ia, ia.data(), (size_t)ma * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mb) HIP_TRY(hipMemcpyAsync(d_ib, ib.data(), (size_t)mb * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_L, &seq_len, 8, hipMemcpyHostToDevice, ctx->stream));
    SimBatch b{};
    b.dense = d_id; b.stride = nn; b.ld = n; b.n = n; b.round_digits = round_digits < 0 ? -1 : round_digits;
    rc = launch_hud_grouped(ctx, b, 1, d_ia, ma, d_ib, mb, oa.empty() ? nullptr : d_oa, ob.empty() ? nullptr : d_ob, threshold,
                            d_L, d_out);
    if (rc) return rc;
    HfstOut o;
    HIP_TRY(hipMemcpyAsync(&o, d_out, sizeof o, hipMemcpyDeviceToHost, ctx->stream));
    rc = ctx_err_fetch(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // ia / ib (pageable) must outlive the copies
    rc = ctx_err_result(ctx, "impop_fst_grouped_from_identity");
    if (rc) return rc;
    for (int k = 0; k < 6; ++k) out[k] = o.v[k];
    if (counts)
        for (int k = 0; k < 6; ++k) counts[k] = o.cnt[k];
    return IMPOP_OK;
}
