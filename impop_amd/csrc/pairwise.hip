// pairwise.hip — the all-pairs path: I_ij = sum_s b_is b_js for every haplotype pair of a
// window, then the full pica2 / h-fst semantics (thresholds, rounding,
// greedy grouping) on identities formed on the fly from the integer Gram matrix.
//
// This path is MFMA-bound, not HBM-bound.  Shipped Gram kernel: gram_fp4_kernel
// (FP4 bit planes, further down); gram_mfma_kernel (int8 + look-up table) is its predecessor,
// selectable with IMPOP_GRAM_MFMA=i8 for A/B measurements.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <type_traits>
#include <vector>

#include "stats_kernels.h"

namespace impop {

// ---- Gram kernel: int8 MFMA, register-only (no LDS, no barriers) ---------------------------
// G[i][j] = sum_s x_i[s] x_j[s] as a dense contraction on the matrix cores: each lane expands the
// bits of ITS rows to 0/1 bytes in registers and feeds v_mfma_i32_32x32x32_i8 (exact: int32
// accumulation).  Design history, all measured on MI355X:
//   * VALU AND + v_bcnt (128x128 tile, 8x8 register tile): v_bcnt_u32_b32 is a HALF-rate
//     instruction on gfx950 (tools/micro/valu_rate.hip: 4.8 vs 2.55 cycles per wave-instruction),
//     ceiling ~1e5 windows/s at n = 465, W = 50 000; reached 23 us/window.
//   * int8 MFMA with bits expanded into LDS byte tiles: ds_write_b128 + one barrier per 64 sites
//     dominate (LDS write path ~79 B/clk/CU); 9 us/window.
//   * int8 MFMA, register-only, rows 1.6 MB apart (plain hap-major): every wave-load touched 32
//     cache lines for 512 useful bytes and the address path, not the ALUs, set the pace; 9-11 us.
//   * 64 x 128 per wave on a ROW-GROUP-BLOCKED operand (RB32, internal.h: one wave-load of a cell
//     for 32 rows is contiguous): 6.4 us/window, MfmaUtil 59 %, VALU ~85 % busy (PMC).
//   * this kernel: 96 x 96 per wave, diagonal tiles reuse A as B, no register copies.
// One WAVE owns one 96 x 96 tile (3 x 3 MFMA tiles of 32 x 32 = 144 accumulator registers), which
// still leaves room for TWO waves per SIMD (a lone wave issues one VALU instruction per ~8
// cycles, two or more reach one per 2.6 / 4.8 cycles).  Lane l supplies row
// (l & 31) of each 32-row group and the 16 sites [16 (l>>5), +16) of a 32-site k-step.  Expansion:
// y = x & 0x0F0F and z = (x >> 4) & 0x0F0F hold the four nibbles as clean bytes, v_mul_u32_u24
// with an SDWA byte select spreads one nibble per instruction (nibble * 0x204081: copies at bits
// 0-3, 7-10, 14-17, 21-24, no carries) and one AND keeps bits 0, 8, 16, 24: 12 VALU ops per
// fragment; 6 fragments per 9 MFMAs (off-diagonal), 3 per 6 (diagonal).  MFMA and VALU runs of one wave do not overlap unless finely
// interleaved (tools/micro/mfma_rate.hip: 257 + 368 -> 589 cycles), hence the software pipeline
// (expand step t+1 under the MFMAs of step t) and the sched_group_barrier issue pattern.
// Both MFMA operands use the same (lane>>5, byte) -> site mapping, so the result does not depend
// on the instruction's internal k order; C/D map: col = lane&31, row = (reg&3)+8(reg>>2)+4(lane>>5).
#ifndef IMPOP_GRAM_ABLATE
#define IMPOP_GRAM_ABLATE 0  // timing-only ablation builds (tools/): bit 0 no lookups, bit 1 no global loads, bit 2 no MFMA
#endif
constexpr int GT = 96;  // tile edge (haplotypes): 3 row groups of 32

struct GramWindow {
    uint64_t site_begin, site_end;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t spread_byte0(uint32_t v, uint32_t k) {
    uint32_t r;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD"
        : "=v"(r) : "v"(v), "v"(k));
    return r;
}
__device__ __forceinline__ uint32_t spread_byte1(uint32_t v, uint32_t k) {
    uint32_t r;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
        : "=v"(r) : "v"(v), "v"(k));
    return r;
}
__device__ __forceinline__ i32x4 expand16(uint32_t xs /* the lane's 16 bits in the low half */, uint32_t kmul) {
    const uint32_t y = xs & 0x0F0Fu;         // nibbles 0, 2 as bytes 0, 1
    const uint32_t z = (xs >> 4) & 0x0F0Fu;  // nibbles 1, 3 as bytes 0, 1
    i32x4 r;
    r.x = (int)(spread_byte0(y, kmul) & 0x01010101u);
    r.y = (int)(spread_byte0(z, kmul) & 0x01010101u);
    r.z = (int)(spread_byte1(y, kmul) & 0x01010101u);
    r.w = (int)(spread_byte1(z, kmul) & 0x01010101u);
    return r;
}

// One task = one 96 x 96 tile pair (ti <= tj) of one window (x one K-slice).  3 x 3 MFMA tiles of
// 32 x 32 per wave (144 accumulators, two waves per SIMD).  Off-diagonal tiles: 6 fragment
// expansions feed 9 MFMAs per 32-site step; diagonal tiles: B == A, so 3 expansions feed the 6
// MFMAs on and above the block diagonal.  Measured motivation (rocprofv3 PMC on the 64 x 128
// predecessor): MfmaUtil 59 %, VALU ~85 % busy, every issue pattern within 1 % => fewer
// expansions per MFMA is the lever (96 also pads 465 haplotypes to 480 instead of 512).
// Pipeline (no register copies): the two 64-site cells PA / PB alternate, fragments F / G alternate;
// a cell is reloaded right after its last use, three steps before its next use.
// Expansion by table: the 8 sites of one byte become 8 int8 through ONE conflict-free ds_read_b64.
// The table is laid out [entry 0..255][lane slot 0..31] x 8 bytes (64 KB per workgroup): a lane only
// ever reads its own slot column, so the 32 lanes of a half-wave always hit 32 different bank
// pairs whatever their entries are.  The LDS address (entry << 8 | slot << 3) is formed by one
// v_perm_b32 of the data dword with a per-lane constant: 1 VALU + 1 LDS read per 8 sites instead of
// 6 VALU, which is what lifts the kernel off the VALU issue limit.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <bool DIAG>
__device__ __forceinline__ void gram_task(const uint32_t *__restrict__ rb, uint64_t nb_row, uint32_t ti, uint32_t tj,
                                          const GramWindow w, uint32_t ks, uint32_t ksplit, int32_t *__restrict__ o,
                                          uint32_t ld, const unsigned char *__restrict__ lut) {
    constexpr int NB = DIAG ? 0 : 3;  // B row groups to load (diagonal: reuse A)
    const uint32_t lane = threadIdx.x & 63, r32 = lane & 31, hi_half = lane >> 5;
    const uint32_t slot = r32 << 3;   // byte 0 of the v_perm source: this lane's 8-byte column in every table row
    i32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    if (w.site_end > w.site_begin) {
        // 32-bit, window-relative indices: cell c = 64 sites = dwords 2c, 2c+1 (relative to the cell of site_begin)
        const uint64_t cell0 = w.site_begin >> 6;
        const uint32_t ncell = (uint32_t)(((w.site_end + 63) >> 6) - cell0);
        const uint32_t cbeg = (uint32_t)((uint64_t)ncell * ks / ksplit);
        const uint32_t cend = (uint32_t)((uint64_t)ncell * (ks + 1) / ksplit);  // this K-slice: cells [cbeg, cend)
        const uint32_t f = (uint32_t)((w.site_begin >> 5) - 2 * cell0);         // first window dword (0 or 1)
        const uint32_t l = (uint32_t)(((w.site_end + 31) >> 5) - 1 - 2 * cell0);  // last window dword
        const uint32_t first_mask = 0xFFFFFFFFu << (w.site_begin & 31);
        const uint32_t last_mask = (w.site_end & 31) ? (0xFFFFFFFFu >> (32 - (w.site_end & 31))) : 0xFFFFFFFFu;
        auto mask_of = [&](uint32_t d) -> uint32_t {  // wave-uniform; zero outside the window AND outside this slice
            uint32_t m = (d >= f && d <= l && d >= 2 * cbeg && d < 2 * cend) ? 0xFFFFFFFFu : 0u;
            if (d == f) m &= first_mask;
            if (d == l) m &= last_mask;
            return m;
        };
        // RB32: dword (row, d) @ (((row>>5) * nb_row + (d>>1)) * 32 + (row&31)) * 2 + (d&1)
        const uint32_t *gA = rb + (((uint64_t)(ti * 3) * nb_row + cell0) * 32 + r32) * 2;
        const uint32_t *gB = rb + (((uint64_t)(tj * 3) * nb_row + cell0) * 32 + r32) * 2;
        const uint64_t g32 = nb_row * 64;  // dwords between consecutive 32-row groups
        // A 64-site cell is two dwords; lane half 0 works on dword 0, lane half 1 on dword 1, and the
        // cell's two 32-site MFMA steps take bytes {0,1} then {2,3} of that dword: every site of the
        // cell is used exactly once, identically for the A and the B operand (the order of sites
        // inside a sum is irrelevant), and no per-lane shift is needed.
        // The window mask of a cell is kept in its own register and applied when the cell is EXPANDED:
        // masking at load time (`load & m`) makes the load's first use immediate and the compiler
        // then waits vmcnt(0) right behind the prefetch (seen in the ISA), which serialises it.
        uint32_t PA_a[3], PB_a[3], PA_b[3], PB_b[3], PA_m, PB_m;
        const uint32_t hsel = hi_half ? 0xFFFFFFFFu : 0u;
        auto load_cell = [&](uint32_t (&ca)[3], uint32_t (&cb)[3], uint32_t &cm, uint32_t c) {  // slack cells keep this in bounds
            const uint32_t m0 = mask_of(2 * c), m1 = mask_of(2 * c + 1);
            cm = (hsel & m1) | (~hsel & m0);  // v_bfi: lane half 1 works on dword 1 (masking A suffices)
#if IMPOP_GRAM_ABLATE & 2  // timing-only build: no global loads
#pragma unroll
            for (int g = 0; g < 3; ++g) ca[g] = c * 2654435761u + g + lane;
#pragma unroll
            for (int g = 0; g < NB; ++g) cb[g] = c * 40503u + g + lane;
            return;
#endif
#pragma unroll
            for (int g = 0; g < 3; ++g) ca[g] = gA[g * g32 + (uint64_t)c * 64 + hi_half];
#pragma unroll
            for (int g = 0; g < NB; ++g) cb[g] = gB[g * g32 + (uint64_t)c * 64 + hi_half];
        };
        i32x4 Fa[3], Fb[3], Ga[3], Gb[3];
        auto lookup16 = [&](uint32_t x, int half) -> i32x4 {  // bytes {2*half, 2*half+1} of x -> 16 int8
#if IMPOP_GRAM_ABLATE & 1  // timing-only build: no v_perm / LDS reads (results are wrong)
            i32x4 q = {(int)x, (int)(x + half), (int)x, (int)x};
            return q;
#endif
            const uint32_t a0 = __builtin_amdgcn_perm(x, slot, half ? 0x0C0C0600u : 0x0C0C0400u);
            const uint32_t a1 = __builtin_amdgcn_perm(x, slot, half ? 0x0C0C0700u : 0x0C0C0500u);
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(lut + a0);
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(lut + a1);
            i32x4 r = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            return r;
        };
        auto expand_step = [&](i32x4 (&fa)[3], i32x4 (&fb)[3], const uint32_t (&ca)[3], const uint32_t (&cb)[3], uint32_t cm,
                               int half) {
#pragma unroll
            for (int g = 0; g < 3; ++g) fa[g] = lookup16(ca[g] & cm, half);
#pragma unroll
            for (int g = 0; g < NB; ++g) fb[g] = lookup16(cb[g], half);
        };
        auto mfma_step = [&](const i32x4 (&fa)[3], const i32x4 (&fb)[3]) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (DIAG && b < a) continue;  // strictly below the block diagonal: by symmetry
#if IMPOP_GRAM_ABLATE & 4  // timing-only build: no MFMA (operands kept alive)
                    asm volatile("" ::"v"(fa[a]), "v"(DIAG ? fa[b] : fb[b]));
                    continue;
#endif
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], DIAG ? fa[b] : fb[b], acc[a][b], 0, 0, 0);
                }
        };
        load_cell(PA_a, PA_b, PA_m, cbeg);
        load_cell(PB_a, PB_b, PB_m, cbeg + 1);
        expand_step(Fa, Fb, PA_a, PA_b, PA_m, 0);
        // Inside a phase: first the address perms and the LDS look-ups of the NEXT step (their latency
        // then hides under this phase's MFMAs, whose own operands were read one phase ago), then the
        // MFMAs; the closing sched_barrier stops any motion across phases.
#define PHASE_ORDER()                                                                      \
    do {                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x002, DIAG ? 9 : 15, 0);  /* VALU: and + perm */ \
        __builtin_amdgcn_sched_group_barrier(0x100, DIAG ? 6 : 12, 0);  /* DS reads */     \
        __builtin_amdgcn_sched_group_barrier(0x008, DIAG ? 6 : 9, 0);   /* MFMA */         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
        // Four phases per iteration; sched_barrier keeps the compiler from pulling a later phase's
        // expansion (and with it the vmcnt wait on the cell just prefetched) forward: without it the
        // ISA showed vmcnt(0) a few instructions behind the loads.  Inside a phase the 12 LDS look-ups
        // of the NEXT step and the 9 MFMAs of the CURRENT step are free to interleave.
        for (uint32_t c = cbeg; c < cend; c += 2) {
            expand_step(Ga, Gb, PA_a, PA_b, PA_m, 1);   // cell c, second half: last use of PA
            load_cell(PA_a, PA_b, PA_m, c + 2);         // reloaded 3 phases before its next use
            mfma_step(Fa, Fb);
            PHASE_ORDER();
            expand_step(Fa, Fb, PB_a, PB_b, PB_m, 0);   // cell c+1 (all-zero A if past the slice)
            mfma_step(Ga, Gb);
            PHASE_ORDER();
            expand_step(Ga, Gb, PB_a, PB_b, PB_m, 1);   // last use of PB
            load_cell(PB_a, PB_b, PB_m, c + 3);
            mfma_step(Fa, Fb);
            PHASE_ORDER();
            expand_step(Fa, Fb, PA_a, PA_b, PA_m, 0);   // cell c+2 for the next iteration
            mfma_step(Ga, Gb);
            PHASE_ORDER();
        }
#undef PHASE_ORDER
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (DIAG && b < a) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t row = ti * GT + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const uint32_t col = tj * GT + 32 * b + r32;
                if (ksplit == 1) o[(uint64_t)row * ld + col] = acc[a][b][e];
                else atomicAdd(&o[(uint64_t)row * ld + col], acc[a][b][e]);
            }
        }
}

// ---- Gram task on the FP4 matrix cores, straight from the raw bit planes (no table, no LDS) ------
// v_mfma_f32_32x32x64_f8f6f4 with E2M1 operands runs K = 64 in the 32 cycles the int8 form needs for
// K = 32 (tools/micro/fp4_probe.hip: 8.45 PFLOP/s over the chip), and a 0/1 matrix needs NO table
// for it: the E2M1 nibble 0010 is 1.0, so the four vectors
//     (x << 1) & 0x22222222,  x & 0x22222222,  (x >> 1) & 0x22222222,  (x >> 2) & 0x22222222
// are valid FP4 operands that together hold every bit of the raw dword x exactly once: 7 VALU per 32
// sites of a row instead of a v_perm + a 64 KB-table look-up per 8 sites.  The four planes of ONE raw
// dword are the four operand dwords of one MFMA (K = 64: lane half 0 supplies 32 sites of cell c,
// lane half 1 the same dword of cell c+1; A and B use the same mapping, so the order of sites inside
// the sum is irrelevant).  fp32 accumulation of 0/1 products is exact below 2^24 (K-slices are
// capped accordingly) and the result is converted to int32 once per task.
// Lane (r = lane & 31, h = lane >> 5) supplies row r of each 32-row group.  RB32: one dwordx2 load of
// a wave = cells c, c+1 of 32 rows = 512 contiguous bytes, and feeds two MFMA phases (dword 0, 1).
// Three cell buffers rotate; a buffer is reloaded right after its last expansion, four phases
// (4 x 9 MFMAs = 1152 cycles) before its next use.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct GramPlanes {          // bit planes of the site weights (weight_planes_kernel), nullptr = unweighted
    const uint32_t *planes;  // plane k at planes + k * stride, one bit per site (dword d = sites 32 d ..)
    uint64_t stride;         // dwords per plane
    uint32_t bits;           // planes with any set bit
};
constexpr uint32_t FP4_MAX_SLICE_PAIRS = (1u << 24) / 128;  // fp32 accumulators stay exact integers
#ifndef IMPOP_GRAM_PACK16
#define IMPOP_GRAM_PACK16 1  // uint16 counts stored two to a dword (0: A/B build with one 2-byte store per count)
#endif
#ifndef IMPOP_GRAM_ORDER
#define IMPOP_GRAM_ORDER 0  // order of a window's tile-pair tasks in its queue (A/B builds: 1, 2; measured equal, profiles/r03_gram_experiments.txt §12)
#endif
#ifndef IMPOP_GRAM_RING
#define IMPOP_GRAM_RING 0  // 0: operands straight into three rotating register cell buffers (shipped); 1: prefetched through a
                          // per-wave LDS ring, 2-3 quads ahead (A/B builds: tools/build_variants.py pairwise.hip ring1:-DIMPOP_GRAM_RING=1).
                          // Measured equal (7.07 vs 6.99 ms per 4096 windows, profiles/r03_gram_experiments.txt): the kernel is not
                          // waiting for its loads, so the simpler form without LDS ships.
#endif
constexpr uint32_t RING_SLOT = 6 * 1024;  // one quad: 6 row groups x (4 cells x 32 rows x 8 B)
constexpr uint32_t RING_SLOTS = 3;
constexpr uint32_t RING_BYTES = RING_SLOT * RING_SLOTS;  // per wave: 18 KB; 4 waves x 2 workgroups = 144 of a CU's 160 KB

__device__ __forceinline__ i32x4 fp4_planes(uint32_t x) {
    i32x4 f;
    f.x = (int)((x << 1) & 0x22222222u);
    f.y = (int)(x & 0x22222222u);
    f.z = (int)((x >> 1) & 0x22222222u);
    f.w = (int)((x >> 2) & 0x22222222u);
    return f;
}
// the builtin takes 8 dwords per operand; FP4 uses the first 4 (the compiler allocates v[n:n+3])
__device__ __forceinline__ i32x8 fp4_operand(const i32x4 f) { return (i32x8){f.x, f.y, f.z, f.w, 0, 0, 0, 0}; }

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A task = tile pair (ti, tj) of a CHAIN of `nchain` windows wins[win0], wins[win0 + wstep], ... (K-slice ks of ksplit; chains
// only with ksplit == 1).  Why chains (round 3): on short windows — a compacted matrix, node-level matrices from a GFA, the
// segments of sliding windows: 2-3 k columns = ~20 pairs = 6 us of MFMAs per task — a task spent more time in its latency chain
// (queue ticket -> window bounds -> first operand loads, each a dependent global round trip) than on the matrix cores.  Inside a
// chain the operand prefetch of the K loop's last iterations reaches into the NEXT window of the chain (same rows, another site
// range: an soffset), and between the weight planes of one window it wraps around to the window's own first pairs, so that only
// the first window of a chain waits for its first loads; one ticket and one scalar load of the bounds per chain link.
template <bool DIAG>
__device__ __forceinline__ void gram_task_fp4(const uint32_t *__restrict__ rb, uint64_t nb_row, uint32_t ti, uint32_t tj,
                                              const GramWindow *__restrict__ wins, uint32_t win0, uint32_t wstep, uint32_t nchain,
                                              uint32_t ks, uint32_t ksplit, int32_t *__restrict__ out, uint64_t out_stride,
                                              uint32_t ld, uint32_t shift, bool add, const GramPlanes wp,
                                              uint32_t ring /* LDS byte address of this wave's operand ring (wave-uniform) */,
                                              bool out16 /* counts stored as uint16 (host: every window's W < 65536, no atomics) */) {
    constexpr int NB = DIAG ? 0 : 3;
    constexpr int NM = DIAG ? 6 : 9;  // MFMAs per phase
    constexpr int NL = DIAG ? 3 : 6;  // row groups = global loads per quad
    const uint32_t lane = threadIdx.x & 63, r32 = lane & 31, hi_half = lane >> 5;
    struct Cell {
        u32x2 a[3], b[3];
    };
    struct Frag {  // four FP4 operand dwords per 32-row group, kept as scalars so that asm can define them singly
        uint32_t a[3][4], b[3][4];
    };
    Cell C0, C1, C2;          // raw cells of three consecutive pairs; across chain links they hold the prefetched first pairs
    bool have_cells = false;  // wave-uniform: C0..C2 already hold pairs 0, 1, 2 of the window (plane pass) about to start
    GramWindow w = wins[win0];
    for (uint32_t link = 0; link < nchain; ++link) {
    const uint32_t win = win0 + link * wstep;
    const bool has_next = link + 1 < nchain;
    const GramWindow wn = has_next ? wins[win + wstep] : w;  // next link's bounds, long before the prefetch needs them
    int32_t *__restrict__ o = out16 ? reinterpret_cast<int32_t *>(reinterpret_cast<uint16_t *>(out) + (uint64_t)win * out_stride)
                                    : out + (uint64_t)win * out_stride;
    uint32_t sh = shift;
    bool stored = false;
    // Weighted sites in ONE task (wp.planes != nullptr): I = sum_k 2^k Gram(M & W_k) by Horner over the used bit planes of
    // the weights, highest first — the K loop below runs once per plane with the plane's words ANDed into the A-side mask,
    // the 144 accumulators are doubled in between (exact: powers of two, totals below 2^24) and written ONCE.  One launch
    // per plane instead (operand masked by a separate kernel, (count << k) added into the output) re-wrote the 480^2
    // counts of every window twelve times: 0.9 ms per plane and 2048 node-level windows, most of it output traffic.
    uint32_t plane_left = wp.planes ? wp.bits : 1u;
    uint32_t kcur = 31u - (uint32_t)__builtin_clz(plane_left);
    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    // entry (row, col) of the tile pair: row = ti*96 + 32a + (e&3) + 8(e>>2) + 4(lane>>5), col = tj*96 + 32b + (lane&31).  The address
    // is a UNIFORM pointer per (a, b, e) (scalar arithmetic) plus ONE per-lane 32-bit offset: 144 per-lane 64-bit addresses would be
    // hoisted out of the chain loop and spilled
    const uint32_t lane_elem = (ti * GT + 4 * (lane >> 5)) * ld + tj * GT + r32;  // < ld^2 <= 2^32 (ld <= 65535 + padding)
    // uint16 counts go out two to a dword: registers e and e + 1 (e even) of a block are the rows r and r + 1 of this lane's
    // column; a lane swaps both with its neighbour column (DPP quad_perm [1,0,3,2]), even lanes then hold (col, col + 1) of row r,
    // odd lanes (col - 1, col) of row r + 1 — one 4-byte store per lane instead of two 2-byte ones (half the store instructions;
    // on short windows a quarter of the launch was its stores, profiles/r03_gram_experiments.txt §10, §13)
    const uint32_t lane_odd = lane & 1u;
    const uint32_t lane_elem2 = lane_elem + (lane_odd ? ld - 1u : 0u);  // row r + 1, column col - 1 for the odd lanes
    auto store_blocks = [&](int a_from, int a_to) {  // the 32 x 32 blocks of accumulator rows [a_from, a_to) -> int32 counts
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (a < a_from || a >= a_to || (DIAG && b < a)) continue;
                if (out16 && IMPOP_GRAM_PACK16) {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const uint64_t eo = (uint64_t)(32 * a + (e & 3) + 8 * (e >> 2)) * ld + 32 * b;  // uniform element offset of (a, b, e)
                        const int32_t v0 = (int32_t)((uint32_t)(int32_t)acc[a][b][e] << sh);
                        const int32_t v1 = (int32_t)((uint32_t)(int32_t)acc[a][b][e + 1] << sh);
                        const int32_t n0 = __builtin_amdgcn_mov_dpp(v0, 0xB1, 0xF, 0xF, true);  // the neighbour column's two rows
                        const int32_t n1 = __builtin_amdgcn_mov_dpp(v1, 0xB1, 0xF, 0xF, true);
                        const uint32_t packed = lane_odd ? (((uint32_t)n1 & 0xFFFFu) | ((uint32_t)v1 << 16))
                                                         : (((uint32_t)v0 & 0xFFFFu) | ((uint32_t)n0 << 16));
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(o) + eo + lane_elem2) = packed;
                    }
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint64_t eo = (uint64_t)(32 * a + (e & 3) + 8 * (e >> 2)) * ld + 32 * b;  // uniform element offset of (a, b, e)
                    int32_t *ob = o + eo;
                    // weighted matrices: this launch is bit plane `shift` of the site weights, added into the other planes' sum
                    const int32_t v = (int32_t)((uint32_t)(int32_t)acc[a][b][e] << sh);
                    if (out16) {  // half the result bytes: a third of a short-window launch is writing 553 KB of counts per window
                        (reinterpret_cast<uint16_t *>(o) + eo)[lane_elem] = (uint16_t)v;
                        continue;
                    }
#if IMPOP_GRAM_ABLATE & 8  // timing-only build: results are not written (only one lane's worth, to keep the work alive)
                    if (lane_elem == 0xFFFFFFFFu) ob[0] = v;
                    continue;
#endif
                    if (ksplit == 1 && !add) ob[lane_elem] = v;
                    else atomicAdd(&ob[lane_elem], v);
                }
            }
    };
    if (w.site_end > w.site_begin) {
        const uint64_t cell0 = w.site_begin >> 6;
        const uint32_t ncell = (uint32_t)(((w.site_end + 63) >> 6) - cell0);
        const uint32_t npair = (ncell + 1) >> 1;  // a pair = cells 2u, 2u+1 (one per lane half) = 128 sites
        const uint32_t ubeg = (uint32_t)((uint64_t)npair * ks / ksplit);
        const uint32_t uend = (uint32_t)((uint64_t)npair * (ks + 1) / ksplit);  // this K-slice: pairs [ubeg, uend)
        const uint32_t f = (uint32_t)((w.site_begin >> 5) - 2 * cell0);           // first window dword (0 or 1)
        const uint32_t l = (uint32_t)(((w.site_end + 31) >> 5) - 1 - 2 * cell0);  // last window dword
        const uint32_t first_mask = 0xFFFFFFFFu << (w.site_begin & 31);
        const uint32_t last_mask = (w.site_end & 31) ? (0xFFFFFFFFu >> (32 - (w.site_end & 31))) : 0xFFFFFFFFu;
        auto mask_of = [&](uint32_t d, bool live) -> uint32_t {  // wave-uniform
            uint32_t m = (live && d >= f && d <= l) ? 0xFFFFFFFFu : 0u;
            if (d == f) m &= first_mask;
            if (d == l) m &= last_mask;
            return m;
        };
        const uint32_t hsel = hi_half ? 0xFFFFFFFFu : 0u;
        auto lane_mask = [&](uint32_t u, int d) -> uint32_t {
            const uint32_t d0 = 4 * u + d;  // dword index of lane half 0 (cell 2u); lane half 1 is one cell (2 dwords) on
            return (hsel & mask_of(d0 + 2, u < uend)) | (~hsel & mask_of(d0, u < uend));
        };
        // the current plane's words for the same two dwords (uniform addresses: scalar loads); past the plane's end the
        // window mask is zero anyway, so the index is only clamped
        const uint32_t *P = nullptr;
        const uint64_t p_lim = wp.planes ? wp.stride - 1 - 2 * cell0 : 0;  // 2 * cell0 < stride: the window starts inside the matrix
        auto plane_mask = [&](uint32_t u, int d) -> uint32_t {
            const uint64_t d0 = 4ull * u + d;
            const uint32_t w0 = P[d0 < p_lim ? d0 : p_lim], w1 = P[d0 + 2 < p_lim ? d0 + 2 : p_lim];
            return (hsel & w1) | (~hsel & w0);
        };
        // Buffer loads: one descriptor per 32-row group, based at the slice's first pair (SGPRs only), the
        // constant per-lane byte offset in voffset and the pair index in soffset: no address VALU at all.
        // (a K-slice is at most FP4_MAX_SLICE_PAIRS * 512 B = 64 MB long, well inside the 32-bit offsets)
        const uint64_t g32b = nb_row * 256;  // bytes between consecutive 32-row groups
        const char *bA = reinterpret_cast<const char *>(rb) + (((uint64_t)(ti * 3) * nb_row + cell0) * 256 + (uint64_t)ubeg * 512);
        const char *bB = reinterpret_cast<const char *>(rb) + (((uint64_t)(tj * 3) * nb_row + cell0) * 256 + (uint64_t)ubeg * 512);
        __amdgpu_buffer_rsrc_t rA[3], rB[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            rA[g] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bA + g * g32b), 0, 0x7FFFFFFF, 0x00020000);
            rB[g] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bB + g * g32b), 0, 0x7FFFFFFF, 0x00020000);
        }
        const uint32_t lane_off = r32 * 8 + hi_half * 256;  // row r32's dword pair inside the lane half's cell
        auto load_one = [&](Cell &C, int i, uint32_t soff) {  // i = 0..2: A groups, 3..5: B groups
            if (i < 3) C.a[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rA[i], lane_off, soff, 0));
            else C.b[i - 3] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rB[i - 3], lane_off, soff, 0));
        };
        // Where the pair loads of the K loop go.  The loop runs n3 = n rounded up to a multiple of 3 pairs (n = uend - ubeg; the
        // filler pairs are fully masked) and loads three pairs ahead, so its last three loads are "pairs" n3, n3 + 1, n3 + 2 into
        // C0, C1, C2: what the NEXT pass starts with — this window's first pairs again if another weight plane follows (`wrap`),
        // else any valid address (the slice's last pair).  The next chain LINK's first pairs are loaded from the store epilogue
        // instead (below): cells alive across all 144 accumulators' stores would not fit the 256 registers of a wave.
        const uint32_t n_pairs = uend - ubeg, n3 = (n_pairs + 2) / 3 * 3;
        bool nx_ok = false;
        uint32_t nx_off = 0, nx_last = 0;  // byte offset of the next link's pair 0 from this slice's descriptor base; its last pair
        if (has_next && ksplit == 1 && wn.site_end > wn.site_begin) {
            const uint64_t cell0n = wn.site_begin >> 6;
            const uint32_t ncelln = (uint32_t)(((wn.site_end + 63) >> 6) - cell0n);
            if (cell0n >= cell0 && (cell0n - cell0) * 256 < 0x7FF00000ull) {
                nx_ok = true;
                nx_off = (uint32_t)((cell0n - cell0) * 256);
                nx_last = ((ncelln + 1) >> 1) - 1;
            }
        }
        bool wrap = false;  // set per plane pass: another plane of this window follows
        auto pair_soff = [&](uint32_t u) -> uint32_t {
            const uint32_t rel = u - ubeg;
            if (rel < n_pairs) return rel * 512u;
            if (rel < n3) return (n_pairs - 1) * 512u;  // masked filler pairs
            const uint32_t j = rel - n3;                // 0, 1, 2
            if (wrap) return (j < n_pairs ? j : n_pairs - 1) * 512u;
            return (n_pairs - 1) * 512u;
        };
        // ---- operand ring in LDS (IMPOP_GRAM_RING = 1, an A/B build; see the macro) -----------------------------------
        // The hypothesis it tested: three cell buffers in registers are a prefetch distance of ~2.5 pairs (1.4 us), less than a
        // loaded HBM / L2-miss latency, and there is no register left for a fourth (2 waves per SIMD x 144 accumulators) — so
        // let the prefetch buffer live in LDS: `buffer_load_dwordx4 ... lds` writes 1 KB per wave instruction — cells c..c+3 of a
        // 32-row group = two pairs, a "quad" — straight into this wave's private ring of RING_SLOTS quads (no VGPR, no barrier:
        // nobody else reads the ring), 2-3 quads = 8-12 phases ahead of use, and a lane fetches its dword pair of the NEXT pair
        // with one ds_read_b64 per row group one phase before the expansion needs it.  The ds_reads are volatile asm (the compiler
        // must not put `vmcnt(0)` in front of LDS reads it would see aliasing the DMA); the waits are explicit:
        //   vmcnt(NL)  before the first ds_read of a quad: everything but the newest quad's NL loads has landed (loads return in
        //              order; the queue pop's atomic drained the previous task's stores),
        //   lgkmcnt(0) before the first expansion that reads a ds_read's destination.
        const uint32_t nquad = (uend - ubeg + 1) >> 1;  // quad j = pairs ubeg + 2j, ubeg + 2j + 1
        auto quad_soff = [&](uint32_t j) -> uint32_t { return (j < nquad ? j : nquad - 1) * 1024u; };  // clamped like pair_soff
        const uint32_t dma_voff = lane * 16;
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        auto dma_one = [&](int i, uint32_t slot_lds, uint32_t soff) {  // i = 0..2: A groups, 3..5: B groups; slot_lds wave-uniform
            if (i < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA[i], (lds_ptr_t)(uintptr_t)(slot_lds + i * 1024), 16, dma_voff, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB[i - 3], (lds_ptr_t)(uintptr_t)(slot_lds + i * 1024), 16, dma_voff, soff, 0, 0);
        };
        const uint32_t lds_lane = ring + r32 * 8 + hi_half * 256;  // this lane's dword pair inside a 512-byte pair of a group
#define DS_CELL(DST, VA, IMM) asm volatile("ds_read_b64 %0, %1 offset:" #IMM : "=v"(DST) : "v"(VA) : "memory")
#define DS_ONE(C, I, VA)                          \
    do {                                          \
        if ((I) == 0) DS_CELL(C.a[0], VA, 0);     \
        if ((I) == 1) DS_CELL(C.a[1], VA, 1024);  \
        if ((I) == 2) DS_CELL(C.a[2], VA, 2048);  \
        if ((I) == 3) DS_CELL(C.b[0], VA, 3072);  \
        if ((I) == 4) DS_CELL(C.b[1], VA, 4096);  \
        if ((I) == 5) DS_CELL(C.b[2], VA, 5120);  \
    } while (0)
        // The phase is scheduled BY HAND in volatile inline asm: builtins let the compiler float the
        // expansion arithmetic across sched_barriers (it is not chained to them) and the MFMAs ended up
        // in runs of 3-9 with the VALU work in one lump behind them.  Volatile asm statements keep their
        // order, so the issue pattern below is the one that runs: MFMA i, then the 4-7 VALU of slot i.
        // Hazards: a fragment register is written >= 7 MFMAs after its last MFMA read and read >= 1 phase
        // after it was written; an accumulator is touched every 9th (6th) MFMA; the compiler cannot see
        // into the asm, so the prologue / epilogue are fenced with explicit s_nop below.
        auto mfma_asm = [](f32x16 &c, const uint32_t (&fa)[4], const uint32_t (&fb)[4]) {
            const i32x4 va = {(int)fa[0], (int)fa[1], (int)fa[2], (int)fa[3]};
            const i32x4 vb = {(int)fb[0], (int)fb[1], (int)fb[2], (int)fb[3]};
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(c) : "v"(va), "v"(vb));
        };
        auto lo_masked = [](uint32_t (&fr)[4], uint32_t &xm, uint32_t x, uint32_t m) {  // planes 0, 1 of x & m (4 VALU)
            asm volatile("v_and_b32 %2, %3, %4\n\tv_lshlrev_b32 %0, 1, %2\n\tv_and_b32 %0, 0x22222222, %0\n\tv_and_b32 %1, 0x22222222, %2"
                         : "=&v"(fr[0]), "=&v"(fr[1]), "=&v"(xm) : "v"(x), "v"(m));
        };
        auto lo_plain = [](uint32_t (&fr)[4], uint32_t x) {  // planes 0, 1 (3 VALU)
            asm volatile("v_lshlrev_b32 %0, 1, %2\n\tv_and_b32 %0, 0x22222222, %0\n\tv_and_b32 %1, 0x22222222, %2"
                         : "=&v"(fr[0]), "=&v"(fr[1]) : "v"(x));
        };
        auto hi_planes = [](uint32_t (&fr)[4], uint32_t x) {  // planes 2, 3 (4 VALU)
            asm volatile("v_lshrrev_b32 %0, 1, %2\n\tv_and_b32 %0, 0x22222222, %0\n\tv_lshrrev_b32 %1, 2, %2\n\tv_and_b32 %1, 0x22222222, %1"
                         : "=&v"(fr[2]), "=&v"(fr[3]) : "v"(x));
        };
#define FP4_PHASE(CURF, NXTF, SRC, D, M, LOADC, SOFF, DO_LOAD)                                            \
    do {                                                                                                  \
        uint32_t xa0 = 0, xa1 = 0, xa2 = 0;                                                               \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                  \
            const int a = DIAG ? (i < 3 ? 0 : i < 5 ? 1 : 2) : i / 3;                                     \
            const int b = DIAG ? (i < 3 ? i : i < 5 ? i - 2 : 2) : i % 3;                                 \
            if (DIAG) mfma_asm(acc[a][b], CURF.a[a], CURF.a[b]);                                          \
            else mfma_asm(acc[a][b], CURF.a[a], CURF.b[b]);                                               \
            if (DO_LOAD && !(IMPOP_GRAM_ABLATE & 2) && i < (DIAG ? 3 : 6)) load_one(LOADC, i, SOFF);      \
            if (IMPOP_GRAM_ABLATE & 1) continue; /* timing-only build: no expansion VALU */               \
            if (DIAG) {                                                                                   \
                if (i == 0) lo_masked(NXTF.a[0], xa0, D ? SRC.a[0].y : SRC.a[0].x, M);                    \
                if (i == 1) hi_planes(NXTF.a[0], xa0);                                                    \
                if (i == 2) lo_masked(NXTF.a[1], xa1, D ? SRC.a[1].y : SRC.a[1].x, M);                    \
                if (i == 3) hi_planes(NXTF.a[1], xa1);                                                    \
                if (i == 4) lo_masked(NXTF.a[2], xa2, D ? SRC.a[2].y : SRC.a[2].x, M);                    \
                if (i == 5) hi_planes(NXTF.a[2], xa2);                                                    \
            } else {                                                                                      \
                if (i == 0) lo_masked(NXTF.a[0], xa0, D ? SRC.a[0].y : SRC.a[0].x, M);                    \
                if (i == 1) { hi_planes(NXTF.a[0], xa0); lo_plain(NXTF.b[0], D ? SRC.b[0].y : SRC.b[0].x); } \
                if (i == 2) hi_planes(NXTF.b[0], D ? SRC.b[0].y : SRC.b[0].x);                            \
                if (i == 3) lo_masked(NXTF.a[1], xa1, D ? SRC.a[1].y : SRC.a[1].x, M);                    \
                if (i == 4) { hi_planes(NXTF.a[1], xa1); lo_plain(NXTF.b[1], D ? SRC.b[1].y : SRC.b[1].x); } \
                if (i == 5) hi_planes(NXTF.b[1], D ? SRC.b[1].y : SRC.b[1].x);                            \
                if (i == 6) lo_masked(NXTF.a[2], xa2, D ? SRC.a[2].y : SRC.a[2].x, M);                    \
                if (i == 7) { hi_planes(NXTF.a[2], xa2); lo_plain(NXTF.b[2], D ? SRC.b[2].y : SRC.b[2].x); } \
                if (i == 8) hi_planes(NXTF.b[2], D ? SRC.b[2].y : SRC.b[2].x);                            \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
        // pair U lives in CUR (its dword 0 is already expanded in F); NXT holds pair U+1.  The mask
        // arithmetic sits behind a wave-uniform branch that only edge pairs take (the empty volatile asm
        // keeps the compiler from turning it back into always-executed selects); the MFMA code is common.
#define FP4_PAIR(CUR, NXT, U)                                                       \
    do {                                                                            \
        uint32_t mA = 0xFFFFFFFFu, mB = 0xFFFFFFFFu;                                \
        if (!(4 * (U) + 1 > f && 4 * (U) + 6 < l && (U) + 1 < uend)) {              \
            asm volatile("");                                                       \
            mA = lane_mask((U), 1);                                                 \
            mB = lane_mask((U) + 1, 0);                                             \
        }                                                                           \
        if (P) {                                                                    \
            mA &= plane_mask((U), 1);                                               \
            mB &= plane_mask((U) + 1, 0);                                           \
        }                                                                           \
        const uint32_t soff = pair_soff((U) + 3);                                   \
        FP4_PHASE(F, G, CUR, 1, mA, CUR, soff, false);                              \
        FP4_PHASE(G, F, NXT, 0, mB, CUR, soff, true);                               \
    } while (0)
        // The ring version of a phase: MFMA i, then (first-half phases) nothing more than the expansion, (second-half phases)
        // the ds_read of row group i of the pair after next into the cell the previous phase finished with; DMA_MODE 1 / 2
        // issues the A / B row groups of the quad three ahead into the slot whose last ds_read has been waited for.
#define FP4_PHASE_R(CURF, NXTF, SRC, D, M, DSC, VA, DO_DS, WAITS, DMA_MODE, SLOT_LDS, SOFF)                  \
    do {                                                                                                  \
        uint32_t xa0 = 0, xa1 = 0, xa2 = 0;                                                               \
        _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                  \
            const int a = DIAG ? (i < 3 ? 0 : i < 5 ? 1 : 2) : i / 3;                                     \
            const int b = DIAG ? (i < 3 ? i : i < 5 ? i - 2 : 2) : i % 3;                                 \
            if (DIAG) mfma_asm(acc[a][b], CURF.a[a], CURF.a[b]);                                          \
            else mfma_asm(acc[a][b], CURF.a[a], CURF.b[b]);                                               \
            if (i == 0 && (WAITS) == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                \
            if (i == 0 && (WAITS) == 2 && DIAG) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");  \
            if (i == 0 && (WAITS) == 2 && !DIAG) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); \
            if (DO_DS && i < NL) DS_ONE(DSC, i, VA);                                                      \
            if (!(IMPOP_GRAM_ABLATE & 2) && (DMA_MODE) == 1 && i < 3) dma_one(i, SLOT_LDS, SOFF);         \
            if (!(IMPOP_GRAM_ABLATE & 2) && (DMA_MODE) == 2 && i < NB) dma_one(3 + i, SLOT_LDS, SOFF);    \
            if (IMPOP_GRAM_ABLATE & 1) continue; /* timing-only build: no expansion VALU */               \
            if (DIAG) {                                                                                   \
                if (i == 0) lo_masked(NXTF.a[0], xa0, D ? SRC.a[0].y : SRC.a[0].x, M);                    \
                if (i == 1) hi_planes(NXTF.a[0], xa0);                                                    \
                if (i == 2) lo_masked(NXTF.a[1], xa1, D ? SRC.a[1].y : SRC.a[1].x, M);                    \
                if (i == 3) hi_planes(NXTF.a[1], xa1);                                                    \
                if (i == 4) lo_masked(NXTF.a[2], xa2, D ? SRC.a[2].y : SRC.a[2].x, M);                    \
                if (i == 5) hi_planes(NXTF.a[2], xa2);                                                    \
            } else {                                                                                      \
                if (i == 0) lo_masked(NXTF.a[0], xa0, D ? SRC.a[0].y : SRC.a[0].x, M);                    \
                if (i == 1) { hi_planes(NXTF.a[0], xa0); lo_plain(NXTF.b[0], D ? SRC.b[0].y : SRC.b[0].x); } \
                if (i == 2) hi_planes(NXTF.b[0], D ? SRC.b[0].y : SRC.b[0].x);                            \
                if (i == 3) lo_masked(NXTF.a[1], xa1, D ? SRC.a[1].y : SRC.a[1].x, M);                    \
                if (i == 4) { hi_planes(NXTF.a[1], xa1); lo_plain(NXTF.b[1], D ? SRC.b[1].y : SRC.b[1].x); } \
                if (i == 5) hi_planes(NXTF.b[1], D ? SRC.b[1].y : SRC.b[1].x);                            \
                if (i == 6) lo_masked(NXTF.a[2], xa2, D ? SRC.a[2].y : SRC.a[2].x, M);                    \
                if (i == 7) { hi_planes(NXTF.a[2], xa2); lo_plain(NXTF.b[2], D ? SRC.b[2].y : SRC.b[2].x); } \
                if (i == 8) hi_planes(NXTF.b[2], D ? SRC.b[2].y : SRC.b[2].x);                            \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
        // pair U lives in CUR (dword 0 already expanded in F), pair U+1 in NXT; the second phase refills CUR with pair U+2 from
        // the ring (address VA).  WAITS 2 on the first pair of a quad (the quad it reads from must have landed), 1 on the second.
#define FP4_PAIR_R(CUR, NXT, U, VA, WAITS, DMA1, DMA2, SLOT_LDS, SOFF)                \
    do {                                                                            \
        uint32_t mA = 0xFFFFFFFFu, mB = 0xFFFFFFFFu;                                \
        if (!(4 * (U) + 1 > f && 4 * (U) + 6 < l && (U) + 1 < uend)) {              \
            asm volatile("");                                                       \
            mA = lane_mask((U), 1);                                                 \
            mB = lane_mask((U) + 1, 0);                                             \
        }                                                                           \
        if (P) {                                                                    \
            mA &= plane_mask((U), 1);                                               \
            mB &= plane_mask((U) + 1, 0);                                           \
        }                                                                           \
        FP4_PHASE_R(F, G, CUR, 1, mA, CUR, VA, false, 0, DMA1, SLOT_LDS, SOFF);     \
        FP4_PHASE_R(G, F, NXT, 0, mB, CUR, VA, true, WAITS, DMA2, SLOT_LDS, SOFF);  \
    } while (0)
        do {  // once, or once per used weight plane (a plain bottom-tested loop: more exits make the compiler shuffle the accumulators)
        if (wp.planes) P = wp.planes + (uint64_t)kcur * wp.stride + 2 * cell0;
        if (ubeg < uend) {
#if IMPOP_GRAM_RING
            Frag F, G;
#pragma unroll
            for (int j = 0; j < (int)RING_SLOTS; ++j)  // quads 0 .. RING_SLOTS-1 on their way
#pragma unroll
                for (int i = 0; i < NL; ++i) dma_one(i < 3 ? i : i, ring + j * RING_SLOT, quad_soff(j));
            if (DIAG) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // quad 0 has landed (RING_SLOTS - 1 quads may still fly)
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            static_assert(RING_SLOTS == 3, "the vmcnt immediates above assume three slots");
            {
                const uint32_t va0 = lds_lane, va1 = lds_lane + 512;
#pragma unroll
                for (int i = 0; i < NL; ++i) { DS_ONE(C0, i, va0); DS_ONE(C1, i, va1); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#else
            Frag F, G;
            wrap = (plane_left & ~(1u << kcur)) != 0;
            if (!have_cells) {  // first link of a chain (or nothing could be prefetched): wait for the first three pairs here
#pragma unroll
                for (int i = 0; i < (DIAG ? 3 : 6); ++i) {
                    load_one(C0, i, pair_soff(ubeg));
                    load_one(C1, i, pair_soff(ubeg + 1));
                    load_one(C2, i, pair_soff(ubeg + 2));
                }
            }
            have_cells = wrap;  // what the loop below leaves in C0..C2 (pair_soff)
#endif
            {
                const uint32_t m0 = lane_mask(ubeg, 0) & (P ? plane_mask(ubeg, 0) : 0xFFFFFFFFu);
                uint32_t xm;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    lo_masked(F.a[g], xm, C0.a[g].x, m0);
                    hi_planes(F.a[g], xm);
                }
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    lo_plain(F.b[g], C0.b[g].x);
                    hi_planes(F.b[g], C0.b[g].x);
                }
#if IMPOP_GRAM_ABLATE & 1  // timing-only build without the in-loop expansion: G needs realistic contents too
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    lo_masked(G.a[g], xm, C1.a[g].y, m0);
                    hi_planes(G.a[g], xm);
                }
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    lo_plain(G.b[g], C1.b[g].y);
                    hi_planes(G.b[g], C1.b[g].y);
                }
#endif
                asm volatile("s_nop 7");  // VALU-written operands -> first MFMA (the compiler cannot see into the asm)
            }
            // no early exit (extra loop exits make the compiler merge 144 accumulators and spill): a slice
            // whose length is not a multiple of 3 pairs runs up to two fully masked pairs
#if IMPOP_GRAM_RING
            // one quad per iteration: its two pairs sit in C0 / C1; quad q + 1 is read out of slot (q + 1) % 3 into the cells
            // as they fall free, quad q + 3 is sent after into slot q % 3 (fully read and waited for one iteration ago)
            uint32_t slot = 0, qj = 0;  // slot of quad q (uniform), quad index relative to the slice
            for (uint32_t u = ubeg; u < uend; u += 2) {
                const uint32_t nslot = slot == RING_SLOTS - 1 ? 0 : slot + 1;
                const uint32_t va = lds_lane + nslot * RING_SLOT;
                const uint32_t slot_lds = ring + slot * RING_SLOT, soff = quad_soff(qj + RING_SLOTS);
                FP4_PAIR_R(C0, C1, u, va, 2, 0, 0, slot_lds, soff);
                FP4_PAIR_R(C1, C0, u + 1, va + 512, 1, 1, 2, slot_lds, soff);
                slot = nslot;
                ++qj;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the cells' last (unused) refills, before the registers are reused
#else
            for (uint32_t u = ubeg; u < uend; u += 3) {
                FP4_PAIR(C0, C1, u);
                FP4_PAIR(C1, C2, u + 1);
                FP4_PAIR(C2, C0, u + 2);
            }
#endif
        }
        asm volatile("s_nop 15\n\ts_nop 15");  // last MFMA results -> the VALU conversions / doublings below
        plane_left &= ~(1u << kcur);
        const uint32_t knext = plane_left ? 31u - (uint32_t)__builtin_clz(plane_left) : kcur;
        const float up = (float)(1u << (kcur - knext));  // planes without a set bit in between are skipped; x1 after the last
        kcur = knext;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (DIAG && b < a) continue;
                acc[a][b] *= up;
            }
        asm volatile("s_nop 7");  // VALU-written accumulators -> the next plane's first MFMA
        } while (plane_left);
        // results out; the next chain link's first three pairs are requested once the first row of accumulator blocks has gone
        // (its registers are free by then), so their latency runs under the other two thirds of the stores and the next link's set-up
        if (wp.planes) sh += kcur;  // Horner stopped at the lowest used plane
        store_blocks(0, 1);
        __builtin_amdgcn_sched_barrier(0);  // the loads stay HERE: hoisted above the stores they would not fit the registers
        if (nx_ok && !(IMPOP_GRAM_RING)) {
#pragma unroll
            for (int i = 0; i < (DIAG ? 3 : 6); ++i) {
                load_one(C0, i, nx_off);
                load_one(C1, i, nx_off + (1 < nx_last ? 1 : nx_last) * 512u);
                load_one(C2, i, nx_off + (2 < nx_last ? 2 : nx_last) * 512u);
            }
            have_cells = true;
        }
        __builtin_amdgcn_sched_barrier(0);
        store_blocks(1, 3);
        stored = true;
#undef FP4_PAIR
#undef FP4_PHASE
#undef FP4_PAIR_R
#undef FP4_PHASE_R
#undef DS_ONE
#undef DS_CELL
    }
    if (!stored) store_blocks(0, 3);  // an empty window never started: zeros
    w = wn;
    }  // chain links
}

// Persistent FP4 Gram kernel: same task queues as gram_mfma_kernel below; LDS only as each wave's private operand ring.
__global__ __launch_bounds__(256, 2) void gram_fp4_kernel(const uint32_t *__restrict__ rb, uint64_t nb_row, uint32_t n_tiles,
                                                          uint32_t tasks_per_win, uint32_t n_win, uint32_t ksplit,
                                                          const GramWindow *__restrict__ wins, int32_t *__restrict__ out,
                                                          uint32_t ld, uint64_t out_stride, uint32_t *__restrict__ queue_heads,
                                                          uint32_t shift, bool add, GramPlanes wp, uint32_t chain, uint32_t out16) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gram_ring[];  // 4 waves x RING_BYTES (dynamic: > 64 KB)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)gram_ring + wave * RING_BYTES;
    const uint32_t slots = tasks_per_win * ksplit;
    const bool by_window = n_win >= 8;
    const uint64_t total = (uint64_t)n_win * slots;
    const uint32_t nc = (by_window && ksplit == 1 && chain > 1) ? chain : 1u;  // windows per ticket (gram_task_fp4: chains)
    for (uint32_t dq = 0; dq < 8; ++dq) {
        const uint32_t q = (blockIdx.x + dq) & 7;
        const uint32_t nwq = (n_win + 7 - q) / 8;  // windows of queue q: q, q + 8, ...
        const uint64_t q_len = by_window ? (uint64_t)((nwq + nc - 1) / nc) * slots : (total + 7 - q) / 8;
        for (;;) {
            uint32_t k = 0;
            if ((threadIdx.x & 63) == 0) k = atomicAdd(&queue_heads[q], 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= q_len) break;
            uint32_t win, t2, links = 1, wstep = 0;
            if (by_window) {
                const uint32_t wb = k / slots;  // ticket = (block of nc windows of this queue, tile pair)
                t2 = k % slots;
                win = q + 8 * (wb * nc);
                links = nwq - wb * nc < nc ? nwq - wb * nc : nc;
                wstep = 8;
            } else { const uint64_t i = q + 8ull * k; win = (uint32_t)(i / slots); t2 = (uint32_t)(i % slots); }
            const uint32_t ks = t2 % ksplit;
            uint32_t rem = t2 / ksplit, ti = 0, tj;
#if IMPOP_GRAM_ORDER == 1  // A/B build (tools/build_variants.py): a window's off-diagonal tile pairs first, its diagonal ones last
            const uint32_t n_off = n_tiles * (n_tiles - 1) / 2;
            if (rem < n_off) {
                while (rem >= n_tiles - 1 - ti) { rem -= n_tiles - 1 - ti; ++ti; }
                tj = ti + 1 + rem;
            } else { ti = tj = rem - n_off; }
#elif IMPOP_GRAM_ORDER == 2  // ... diagonal ones first
            if (rem < n_tiles) { ti = tj = rem; }
            else {
                rem -= n_tiles;
                while (rem >= n_tiles - 1 - ti) { rem -= n_tiles - 1 - ti; ++ti; }
                tj = ti + 1 + rem;
            }
#else
            while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ++ti; }
            tj = ti + rem;
#endif
            if (ti == tj) gram_task_fp4<true>(rb, nb_row, ti, tj, wins, win, wstep, links, ks, ksplit, out, out_stride, ld, shift, add, wp, ring, out16 != 0);
            else gram_task_fp4<false>(rb, nb_row, ti, tj, wins, win, wstep, links, ks, ksplit, out, out_stride, ld, shift, add, wp, ring, out16 != 0);
        }
    }
}

// Persistent workgroups (2 per CU, 4 waves each) that share only the lookup table; every WAVE pulls
// (window, tile pair, K-slice) tasks from one of 8 queues until all are drained, so a wave whose task
// was short (diagonal tiles do 2/3 of the MFMAs) immediately starts another one and both SIMD slots
// stay occupied (PMC before this change: 1.45-1.6 resident waves per SIMD, after: ~2).
// Queue q holds the tasks of windows with win % 8 == q and is served first by workgroups with
// blockIdx % 8 == q, i.e. (under the observed round-robin placement) by one XCD, whose L2 then holds
// that window's rows for all of its 15 tile pairs; a workgroup whose queue is empty steals from the
// others, so the result and termination never depend on placement: every wave leaves once all eight
// counters have passed their queue length.
// ksplit > 1: the site range of a window is cut into ksplit slices handled by different tasks that
// atomicAdd into a zero-initialised output (integer adds commute: still bit-reproducible); used
// when there are too few (window, tile) tasks to keep two waves on every SIMD.
__global__ __launch_bounds__(256, 2) void gram_mfma_kernel(const uint32_t *__restrict__ rb, uint64_t nb_row,
                                                           uint32_t n_tiles, uint32_t tasks_per_win, uint32_t n_win,
                                                           uint32_t ksplit, const GramWindow *__restrict__ wins,
                                                           int32_t *__restrict__ out, uint32_t ld, uint64_t out_stride,
                                                           uint32_t *__restrict__ queue_heads /* 8, zeroed per launch */) {
    __shared__ __attribute__((aligned(16))) unsigned char lut[256 * 32 * 8];  // 64 KB: [entry][lane slot] x 8 bytes
    for (uint32_t i = threadIdx.x; i < 256 * 32; i += 256) {
        const uint32_t e = i >> 5;  // entry: bit j of e -> byte j
        u32x2 v;
        v.x = __umul24(e & 0xFu, 0x204081u) & 0x01010101u;
        v.y = __umul24((e >> 4) & 0xFu, 0x204081u) & 0x01010101u;
        *reinterpret_cast<u32x2 *>(lut + (size_t)i * 8) = v;
    }
    __syncthreads();  // the only barrier; everything below is per wave
    const uint32_t slots = tasks_per_win * ksplit;  // (tile pair, K-slice) slots of one window
    const bool by_window = n_win >= 8;              // few windows: deal single tasks round-robin instead
    const uint64_t total = (uint64_t)n_win * slots;
    for (uint32_t dq = 0; dq < 8; ++dq) {
        const uint32_t q = (blockIdx.x + dq) & 7;
        const uint64_t q_len = by_window ? (uint64_t)((n_win + 7 - q) / 8) * slots : (total + 7 - q) / 8;
        for (;;) {
            uint32_t k = 0;
            if ((threadIdx.x & 63) == 0) k = atomicAdd(&queue_heads[q], 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= q_len) break;  // queue drained (the head keeps counting, harmlessly)
            uint32_t win, t2;
            if (by_window) { win = q + 8 * (k / slots); t2 = k % slots; }
            else { const uint64_t i = q + 8ull * k; win = (uint32_t)(i / slots); t2 = (uint32_t)(i % slots); }
            const uint32_t ks = t2 % ksplit;
            uint32_t rem = t2 / ksplit, ti = 0;
            while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ++ti; }
            const uint32_t tj = ti + rem;
            int32_t *o = out + (uint64_t)win * out_stride;
            if (ti == tj) gram_task<true>(rb, nb_row, ti, tj, wins[win], ks, ksplit, o, ld, lut);
            else gram_task<false>(rb, nb_row, ti, tj, wins[win], ks, ksplit, o, ld, lut);
        }
    }
}

// The operand is stored in minor-allele polarity (layout.hip sb_to_hm_kernel; m->phi_row): the counts the Gram kernel wrote are
// I'_ij of the stored bits, with row / column p = phi_row holding I'_ip = (complemented sites haplotype i is stored with a 1 at)
// and I'_pp = the number of complemented sites (weights: their summed weights).  For a complemented site b = 1 - b', so
//     I_ij = I'_ij + I'_pp - I'_ip - I'_jp          (i <= j < n; the diagonal a_i = I_ii likewise)
// Exact integers.  Needed where I or a themselves matter — the `dice` identity, exported counts; the `match` identity and every
// statistic built on it see only a_i + a_j - 2 I_ij, which is the same in both polarities, and skip this pass.
// grid: (matrices, row chunks); dynamic LDS: n int32.
template <typename T>  // int32_t, or uint16_t (counts of short windows, SimBatch.g16)
__global__ __launch_bounds__(256) void gram_unflip_kernel(T *__restrict__ g, uint32_t ld, uint64_t stride, uint32_t n, uint32_t phi) {
    extern __shared__ int32_t unflip_t[];
    T *G = g + (uint64_t)blockIdx.x * stride;
    for (uint32_t i = threadIdx.x; i < n; i += 256) unflip_t[i] = (int32_t)G[(uint64_t)i * ld + phi];
    const int32_t P = (int32_t)G[(uint64_t)phi * ld + phi];
    __syncthreads();
    for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
        const int32_t ri = P - unflip_t[i];
        T *row = G + (uint64_t)i * ld;
        for (uint32_t j = i + threadIdx.x; j < n; j += 256) row[j] = (T)((int32_t)row[j] + ri - unflip_t[j]);  // column phi itself (j = n) stays as written
    }
}
static int launch_gram_unflip(impop_ctx *ctx, const impop_matrix *m, int32_t *d_g, uint64_t n_mats, bool g16 = false) {
    if (m->phi_row == 0xFFFFFFFFu || n_mats == 0) return IMPOP_OK;
    const uint32_t n = m->g.n_hap, ld = m->n_hap_pad;
    REQUIRE(n_mats < 0x7FFFFFFFull && (size_t)n * 4 <= 64 * 1024, "gram_unflip: too many matrices / haplotypes");
    const uint32_t chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096 / n_mats, 1), 64);
    if (g16)
        hipLaunchKernelGGL(gram_unflip_kernel<uint16_t>, dim3((uint32_t)n_mats, chunks), dim3(256), (size_t)n * 4, ctx->stream,
                           reinterpret_cast<uint16_t *>(d_g), ld, (uint64_t)ld * ld, n, m->phi_row);
    else
        hipLaunchKernelGGL(gram_unflip_kernel<int32_t>, dim3((uint32_t)n_mats, chunks), dim3(256), (size_t)n * 4, ctx->stream, d_g, ld,
                           (uint64_t)ld * ld, n, m->phi_row);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

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
                                         const double *__restrict__ taj /* a1,a2,b1,b2,c1,c2,e1,e2 for n = nP */,
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
        TajConsts c;
        c.a1 = taj[0]; c.a2 = taj[1]; c.b1 = taj[2]; c.b2 = taj[3]; c.c1 = taj[4]; c.c2 = taj[5]; c.e1 = taj[6]; c.e2 = taj[7];
        D = tajima_d_from(c, S, pin, nullptr, nullptr);
    }
    r.tajima_d = D;
    out[i] = r;
}

// IMPOP_GRAM_MFMA=i8 selects the int8 kernel (kept for A/B measurements); default: FP4 bit planes
static bool gram_use_fp4() {
    static const bool v = [] {
        const char *e = getenv("IMPOP_GRAM_MFMA");
        return !(e && e[0] == 'i');
    }();
    return v;
}

// add_shift < 0: d_out = Gram; >= 0 (FP4 kernel only): d_out += Gram << add_shift (d_out holds the other planes' sum).
// fused_planes (FP4 kernel only): the weighted Gram matrix of m in one launch (gram_task_fp4), every window's summed
// weight below 2^24.
// out16 (in / out, nullable): the caller would take uint16 counts (every window's W < 65536); set to whether the launch wrote them
// (only the FP4 kernel, only without K-split atomics and plane accumulation)
static int launch_gram(impop_ctx *ctx, const impop_matrix *m, const uint32_t *d_rb, const GramWindow *d_wins, uint32_t n_win,
                       int32_t *d_out, uint64_t max_window_sites, int add_shift = -1, bool fused_planes = false, bool *out16 = nullptr) {
    const uint32_t T = m->n_hap_pad / GT;
    const uint32_t tasks_per_win = T * (T + 1) / 2;  // upper-triangular tile pairs
    // two waves per SIMD on every CU = 8 * n_cu resident waves; aim at >= 4 rounds of them so the
    // last, partially filled round does not dominate, and split the site axis when there are fewer tasks
    const uint64_t want = 32ull * (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    uint32_t ksplit = 1;
    while ((uint64_t)n_win * tasks_per_win * ksplit < want && ksplit < 64) ksplit *= 2;
    // FP4: fp32 accumulators must stay below 2^24 per K-slice
    while (gram_use_fp4() && (max_window_sites / 128 + 2) / ksplit + 1 > FP4_MAX_SLICE_PAIRS) ksplit *= 2;
    const bool w16 = out16 && *out16 && gram_use_fp4() && ksplit == 1 && add_shift < 0;
    if (out16) *out16 = w16;
    if (ksplit > 1 && add_shift < 0)
        HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)n_win * m->n_hap_pad * m->n_hap_pad * sizeof(int32_t), ctx->stream));
    REQUIRE((uint64_t)n_win * tasks_per_win * ksplit < 0xFFFFFFF0ull, "gram: too many tasks for one launch");
    if (!ctx->d_queue) HIP_TRY(hipMalloc((void **)&ctx->d_queue, 8 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(ctx->d_queue, 0, 8 * sizeof(uint32_t), ctx->stream));
    // persistent grid: two 256-thread workgroups per CU (64 KB of LDS each), never fewer than 8
    const uint32_t n_cu = (uint32_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    const uint64_t need_wg = ((uint64_t)n_win * tasks_per_win * ksplit + 3) / 4;
    // IMPOP_GRAM_WG_PER_CU=1 (A/B runs): one workgroup = one wave per SIMD on every CU (the LDS request keeps a second one out)
    static const uint32_t wg_per_cu = [] { const char *e = getenv("IMPOP_GRAM_WG_PER_CU"); return (e && e[0] == '1') ? 1u : 2u; }();
    const uint32_t grid = (uint32_t)std::max<uint64_t>(8, std::min<uint64_t>((uint64_t)wg_per_cu * n_cu, need_wg));
    const uint32_t ring_lds = wg_per_cu == 1 ? 100u * 1024u : (IMPOP_GRAM_RING ? 4 * RING_BYTES : 0u);
    GramPlanes wp{nullptr, 0, 0};
    if (fused_planes) wp = GramPlanes{m->d_wplanes, m->wplane_stride, m->wplane_bits};
    // chains of windows per ticket (gram_task_fp4) where a task is short — at most 8192 columns = 64 pairs, 20 us of MFMAs — and
    // there are enough tasks that every wave still draws >= 4 tickets (so the grid's last round stays small); IMPOP_GRAM_CHAIN=n overrides
    uint32_t chain = 1;
    const uint64_t planes_walked = fused_planes ? std::max<uint32_t>((uint32_t)__builtin_popcount(m->wplane_bits), 1u) : 1u;  // K passes per window
    if (ksplit == 1 && n_win >= 8 && max_window_sites * planes_walked <= 8192) {
        const uint64_t per_wave = (uint64_t)n_win * tasks_per_win / (8ull * n_cu);
        chain = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(per_wave / 4, 1), 8);  // measured (tools/ab_chain.py): 1.90 / 1.79 / 1.75 ms
    }                                                                                  // per 4096 compacted windows at 1 / 3 / 8 links
    {
        static const int forced = [] { const char *e = getenv("IMPOP_GRAM_CHAIN"); return e ? atoi(e) : 0; }();
        if (forced > 0 && ksplit == 1) chain = (uint32_t)forced;
    }
    {
        static const hipError_t ring_attr = hipFuncSetAttribute((const void *)gram_fp4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                (int)(100 * 1024));  // 72 KB per workgroup: above the 64 KB default
        HIP_TRY(ring_attr);
    }
    if (gram_use_fp4())
        hipLaunchKernelGGL(gram_fp4_kernel, dim3(grid), dim3(256), ring_lds, ctx->stream, d_rb, m->rb_nb, T, tasks_per_win, n_win,
                           ksplit, d_wins, d_out, m->n_hap_pad, (uint64_t)m->n_hap_pad * m->n_hap_pad, ctx->d_queue,
                           add_shift < 0 ? 0u : (uint32_t)add_shift, add_shift >= 0, wp, chain, w16 ? 1u : 0u);
    else
        hipLaunchKernelGGL(gram_mfma_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_rb, m->rb_nb, T, tasks_per_win, n_win,
                           ksplit, d_wins, d_out, m->n_hap_pad, (uint64_t)m->n_hap_pad * m->n_hap_pad, ctx->d_queue);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

// every window lighter than 2^24 (fp32-exact sums): all weight planes inside one launch (Horner in gram_task_fp4);
// IMPOP_GRAM_PLANES=split keeps the launch-per-plane form for A/B measurements
constexpr uint64_t GRAM_FUSED_WEIGHT_LIMIT = 1ull << 24;
static bool gram_planes_in_task() {
    static const bool split_planes = [] { const char *e = getenv("IMPOP_GRAM_PLANES"); return e && e[0] == 's'; }();
    return gram_use_fp4() && !split_planes;
}

// ---- weighted sites on the all-pairs path ------------------------------------------------------------------
// Column s stands for w_s base pairs (one column per graph node, impop_matrix_set_site_weights): what `impg
// similarity` hands the reference is a bp-weighted node-sharing identity (run_pica2_impg.sh:162-175), i.e.
//     I_ij = sum_s w_s b_is b_js.
// Write w_s = sum_k 2^k w_ks with bit planes w_ks in {0,1}.  Masking the site axis with plane k gives a 0/1
// matrix M_k = M & W_k whose plain Gram matrix is sum_s w_ks b_is b_js (w_ks^2 = w_ks), so
//     I = sum_k 2^k Gram(M_k)
// with the same matrix-core pipeline.  Usual case (every window lighter than 2^24): ONE launch, the planes walked inside
// each task (gram_task_fp4, GramPlanes).  Otherwise, per plane, one elementwise AND of the RB32 operand (only the cells
// the batch touches) and one Gram launch whose epilogue ADDS (count << k) into the sum of the planes before it (integer
// atomics; a separate shifted-accumulate pass over 4096 x 480^2 counts cost 1.0 ms per plane, twice the Gram launch it
// followed).  tools/bench_weighted.py, 4096 node-level windows of 12 planes: 46 ms (separate pass) -> 30 ms (epilogue
// adds) -> 8.3 ms (planes in the task).  Exact by construction (every partial Gram is an exact integer; the sum is required
// to stay below 2^31 like the length of an unweighted window); planes without any set bit are skipped, so node lengths
// below 2^p cost p passes over the NODE-level matrix — against mean-node-length times the MACs for the bp-expanded one.
__global__ void weight_planes_kernel(const uint32_t *__restrict__ wt, uint64_t n_site, uint64_t n_dword, uint32_t n_plane,
                                     uint32_t *__restrict__ planes, uint32_t *__restrict__ used_bits) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dword) return;
    uint32_t any = 0;
    uint32_t w[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const uint64_t s = 32 * d + j;
        w[j] = s < n_site ? wt[s] : 0u;
        any |= w[j];
    }
    for (uint32_t k = 0; k < n_plane; ++k) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) bits |= ((w[j] >> k) & 1u) << j;
        planes[(uint64_t)k * n_dword + d] = bits;
    }
    if (any) atomicOr(used_bits, any);
}

// masked operand for plane k over the cells [cell_lo, cell_hi) of every 32-row group (one thread = one row's dword pair)
__global__ void rb_mask_kernel(const uint32_t *__restrict__ rb, uint32_t *__restrict__ out, uint64_t rb_nb, uint32_t n_group,
                               uint64_t cell_lo, uint64_t cell_hi, const uint32_t *__restrict__ plane, uint64_t n_dword) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_group = (cell_hi - cell_lo) * 32;
    if (t >= per_group * n_group) return;
    const uint64_t g = t / per_group, r = t % per_group, cell = cell_lo + r / 32;
    const uint64_t at = ((g * rb_nb + cell) * 32 + (r & 31)) * 2;
    const u32x2 v = *reinterpret_cast<const u32x2 *>(rb + at);
    u32x2 o;
    o.x = 2 * cell < n_dword ? v.x & plane[2 * cell] : 0u;
    o.y = 2 * cell + 1 < n_dword ? v.y & plane[2 * cell + 1] : 0u;
    *reinterpret_cast<u32x2 *>(out + at) = o;
}

__global__ void gram_accumulate_kernel(int32_t *__restrict__ acc, const int32_t *__restrict__ part, uint32_t shift, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) acc[i] += (int32_t)((uint32_t)part[i] << shift);
}

static int ensure_weight_planes(impop_ctx *ctx, const impop_matrix *m) {
    if (m->d_wplanes) return IMPOP_OK;
    const uint64_t n_dword = 2 * m->g.n_block;
    if (n_dword == 0) {  // no column at all (a weighted matrix compacted to nothing): no plane has a bit
        m->wplane_bits = 0;
        m->wplane_stride = 0;
        return IMPOP_OK;
    }
    uint32_t *d_used = nullptr;
    HIP_TRY(hipMalloc((void **)&m->d_wplanes, 32ull * n_dword * 4 + 256));
    d_used = m->d_wplanes + 32ull * n_dword;
    HIP_TRY(hipMemsetAsync(d_used, 0, 4, ctx->stream));
    hipLaunchKernelGGL(weight_planes_kernel, dim3((uint32_t)((n_dword + 127) / 128)), dim3(128), 0, ctx->stream, m->d_wt, m->g.n_site,
                       n_dword, 32u, m->d_wplanes, d_used);
    HIP_TRY(hipGetLastError());
    uint32_t used = 0;
    HIP_TRY(hipMemcpyAsync(&used, d_used, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    m->wplane_bits = used;
    m->wplane_stride = n_dword;
    if (!m->d_rb_masked) HIP_TRY(hipMalloc((void **)&m->d_rb_masked, m->rb_bytes));
    return IMPOP_OK;
}

// Gram matrices of `n_win` cells (host copy h_wins of d_wins for the cell range) into d_out; d_tmp: a second buffer
// of the same size, used by weighted matrices only
static int launch_gram_any(impop_ctx *ctx, const impop_matrix *m, const GramWindow *d_wins, const GramWindow *h_wins,
                           uint32_t n_win, int32_t *d_out, int32_t *d_tmp, uint64_t max_window_sites, bool *out16 = nullptr) {
    if (m->wt_prefix.empty()) return launch_gram(ctx, m, m->d_rb, d_wins, n_win, d_out, max_window_sites, -1, false, out16);
    int rc = ensure_weight_planes(ctx, m);
    if (rc) return rc;
    uint64_t c_lo = ~0ull, c_hi = 0;
    for (uint32_t i = 0; i < n_win; ++i)
        if (h_wins[i].site_end > h_wins[i].site_begin) {
            c_lo = std::min(c_lo, h_wins[i].site_begin >> 6);
            c_hi = std::max(c_hi, (h_wins[i].site_end + 63) >> 6);
        }
    const uint64_t count = (uint64_t)n_win * m->n_hap_pad * m->n_hap_pad;
    uint64_t heaviest = 0;
    const std::vector<uint64_t> &pre = m->compact ? m->kept_wt_prefix : m->wt_prefix;  // the cells are in MATRIX coordinates
    for (uint32_t i = 0; i < n_win; ++i)
        if (h_wins[i].site_end > h_wins[i].site_begin)
            heaviest = std::max(heaviest, pre[h_wins[i].site_end] - pre[h_wins[i].site_begin]);
    if (gram_planes_in_task() && m->wplane_bits && heaviest < GRAM_FUSED_WEIGHT_LIMIT) {
        if (out16 && heaviest >= 65536) *out16 = false;
        return launch_gram(ctx, m, m->d_rb, d_wins, n_win, d_out, max_window_sites, -1, true, out16);
    }
    if (out16) *out16 = false;  // one launch per plane, accumulated with atomics: 32-bit counts
    REQUIRE(d_tmp || !m->wplane_bits, "weighted Gram: no buffer for the plane partials");
    HIP_TRY(hipMemsetAsync(d_out, 0, count * 4, ctx->stream));
    if (c_lo >= c_hi) return IMPOP_OK;
    c_hi = std::min<uint64_t>(c_hi + 8, m->rb_nb);  // the Gram pipeline prefetches a few cells past a window's end
    const uint32_t n_group = m->n_hap_pad / 32;
    const uint64_t threads = (c_hi - c_lo) * 32 * n_group;
    REQUIRE((threads + 255) / 256 < 0x7FFFFFFFull && (count + 255) / 256 < 0x7FFFFFFFull, "weighted Gram: batch too large");
    for (uint32_t k = 0; k < 32; ++k) {
        if (!((m->wplane_bits >> k) & 1u)) continue;
        hipLaunchKernelGGL(rb_mask_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream, m->d_rb, m->d_rb_masked,
                           m->rb_nb, n_group, c_lo, c_hi, m->d_wplanes + (uint64_t)k * m->wplane_stride, m->wplane_stride);
        HIP_TRY(hipGetLastError());
        if (gram_use_fp4()) {  // the plane's shift and the sum over planes happen in the Gram kernel's own stores
            rc = launch_gram(ctx, m, m->d_rb_masked, d_wins, n_win, d_out, max_window_sites, (int)k);
            if (rc) return rc;
            continue;
        }
        rc = launch_gram(ctx, m, m->d_rb_masked, d_wins, n_win, d_tmp, max_window_sites);
        if (rc) return rc;
        hipLaunchKernelGGL(gram_accumulate_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_out, d_tmp, k, count);
        HIP_TRY(hipGetLastError());
    }
    return IMPOP_OK;
}

// ---- S for the all-pairs path from a cached site bitmap -----------------------------------------------------------
// S = #{s in window : 0 < c_s < n} (all haplotypes: run_tajd.sh:126,148 take S from the un-subset graph).  Whether a
// site segregates does not depend on the window, so it is computed ONCE per matrix — one streaming pass over the
// SB64 layout, one bit per site — and kept with the matrix; a window's S is then a popcount over W / 8 bytes
// instead of a second pass over its n W / 8 bytes behind every Gram launch (that pass was 1.83 ms of every
// 12 ms batch of 4096 windows; the first call on a matrix still pays it once).
typedef uint32_t u32q __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void segmap_kernel(const uint32_t *__restrict__ sb, uint32_t wps, uint32_t G, uint32_t r, uint32_t n,
                                                     uint64_t n_block, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_block) return;  // wave-uniform
    const uint32_t *blk = sb + b * 64ull * wps;
    uint32_t c = 0;
    for (uint32_t g = 0; g + 1 < G; ++g) {
        const u32q v = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(blk + (uint64_t)g * 256 + lane * 4));
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (uint32_t e = 0; e < r; ++e) c += __popc(__builtin_nontemporal_load(blk + (uint64_t)(G - 1) * 256 + lane * r + e));
    const uint64_t bal = __ballot((c - 1u) < (n - 1u));  // padding sites of the last block are all-zero: not segregating
    if (lane == 0) { out[2 * b] = (uint32_t)bal; out[2 * b + 1] = (uint32_t)(bal >> 32); }
}

// one wave per window: popcount of the bitmap over [site_begin, site_end) -> s_all and s_p of the window's record
__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t *__restrict__ map, const GramWindow *__restrict__ wins,
                                                        uint64_t n_win, impop_window_stats *__restrict__ stats /* nullable */,
                                                        uint32_t *__restrict__ plain /* nullable: one count per window */) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_win) return;
    const uint64_t s0 = wins[w].site_begin, s1 = wins[w].site_end;
    const uint64_t d0 = s0 >> 5, d1 = (s1 + 31) >> 5;
    uint32_t cnt = 0;
    for (uint64_t d = d0 + lane; d < d1; d += 64) {
        uint32_t v = map[d];
        if (d == d0) v &= 0xFFFFFFFFu << (s0 & 31);
        if (d == d1 - 1 && (s1 & 31)) v &= 0xFFFFFFFFu >> (32 - (s1 & 31));
        cnt += __popc(v);
    }
    cnt = wave_sum_u32(cnt);
    if (lane == 0) {
        if (stats) { stats[w].s_all = cnt; stats[w].s_p = cnt; }
        if (plain) plain[w] = cnt;
    }
}

// compacted matrix: the dropped all-ones sites of [site_begin, site_end) (ORIGINAL coordinates) add 1 to every I_ij
__global__ void gram_add_const_kernel(int32_t *__restrict__ g, uint32_t ld, uint32_t n, const uint32_t *__restrict__ add) {
    const uint32_t i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && j < n) g[(uint64_t)i * ld + j] += (int32_t)add[0];
}

// kept-site index range of an original-coordinate range on a compacted matrix (identity otherwise)
static inline void map_range(const impop_matrix *m, uint64_t s0, uint64_t s1, uint64_t *k0, uint64_t *k1) {
    if (!m->compact) { *k0 = s0; *k1 = s1; return; }
    *k0 = pos_lower_bound(m, s0);
    *k1 = pos_lower_bound(m, s1);
}

int ensure_segmap(impop_ctx *ctx, const impop_matrix *m) {
    if (m->d_segmap || m->g.n_block == 0) return IMPOP_OK;
    HIP_TRY(hipMalloc((void **)&m->d_segmap, m->g.n_block * 8 + 256));
    REQUIRE((m->g.n_block + 3) / 4 < 0x7FFFFFFFull, "site bitmap: matrix too long for one launch");
    hipLaunchKernelGGL(segmap_kernel, dim3((uint32_t)((m->g.n_block + 3) / 4)), dim3(256), 0, ctx->stream, m->d_sb, m->g.wps, m->g.G,
                       m->g.r, m->g.n_hap, m->g.n_block, m->d_segmap);
    HIP_TRY(hipGetLastError());
    return IMPOP_OK;
}

// W of a window: its length, or the sum of its columns' weights
static inline uint64_t window_W(const impop_matrix *m, uint64_t s0, uint64_t s1) {
    return m->wt_prefix.empty() ? s1 - s0 : m->wt_prefix[s1] - m->wt_prefix[s0];
}

// compacted from a weighted matrix: the summed weights of the dropped all-ones sites of [s0, s1) (original coordinates)
static inline uint32_t ones_weight(const impop_matrix *m, uint64_t s0, uint64_t s1) {
    return (uint32_t)(m->ones_wt_prefix[s1] - m->ones_wt_prefix[s0]);  // < 2^31: part of the window's W
}
static inline bool compact_weighted(const impop_matrix *m) { return m->compact && !m->ones_wt_prefix.empty(); }

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
    if (m->compact && !(m->d_rb && m->d_onesmap)) {
        set_error("%s: this compacted matrix has no all-pairs operand (compact a matrix that kept IMPOP_KEEP_HAP_MAJOR)", fn);
        return IMPOP_E_UNSUPPORTED;
    }
    REQUIRE(m->d_rb, "%s: matrix was created without IMPOP_KEEP_HAP_MAJOR", fn);
    REQUIRE(s0 <= s1 && s1 <= matrix_span(m), "%s: bad site range [%llu,%llu)", fn, (unsigned long long)s0,
            (unsigned long long)s1);
    REQUIRE(window_W(m, s0, s1) < (1ull << 31), "%s: window of 2^31 or more sites (or summed site weights) overflows int32 counts", fn);
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
    rc = ctx_scratch(ctx, 2048 + (size_t)ld * ld * 8, &d);
    if (rc) return rc;
    Carve2 cv(d);
    GramWindow *d_w = cv.take<GramWindow>(1);
    int32_t *d_g = cv.take<int32_t>((size_t)ld * ld);
    int32_t *d_t = cv.take<int32_t>((size_t)ld * ld);
    GramWindow *d_ow = cv.take<GramWindow>(1);
    uint32_t *d_add = cv.take<uint32_t>(1);
    GramWindow w;
    map_range(m, site_begin, site_end, &w.site_begin, &w.site_end);  // compacted: the kept sites of the range
    HIP_TRY(hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gram_any(ctx, m, d_w, &w, 1, d_g, d_t, w.site_end - w.site_begin);
    if (rc) return rc;
    rc = launch_gram_unflip(ctx, m, d_g, 1);  // exported counts / identities: the original polarity
    if (rc) return rc;
    const GramWindow ow{site_begin, site_end};
    const uint32_t add_w = compact_weighted(m) ? ones_weight(m, site_begin, site_end) : 0u;
    if (m->compact) {  // + the dropped sites every haplotype carries (their count, or their summed weights)
        if (compact_weighted(m)) {
            HIP_TRY(hipMemcpyAsync(d_add, &add_w, 4, hipMemcpyHostToDevice, ctx->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(d_ow, &ow, sizeof ow, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(seg_count_kernel, dim3(1), dim3(256), 0, ctx->stream, m->d_onesmap, d_ow, 1, (impop_window_stats *)nullptr, d_add);
        }
        hipLaunchKernelGGL(gram_add_const_kernel, dim3((n + 15) / 16, (n + 15) / 16), dim3(16, 16), 0, ctx->stream, d_g, ld, n, d_add);
        HIP_TRY(hipGetLastError());
    }
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
    rc = ctx_scratch(ctx, 4096 + (size_t)ld * ld * 8 + (size_t)n * n * 8, &d);
    if (rc) return rc;
    Carve2 cv(d);
    GramWindow *d_w = cv.take<GramWindow>(1);
    uint64_t *d_W = cv.take<uint64_t>(1);
    int32_t *d_g = cv.take<int32_t>((size_t)ld * ld);
    int32_t *d_t = cv.take<int32_t>((size_t)ld * ld);
    double *d_id = cv.take<double>((size_t)n * n);
    GramWindow *d_ow = cv.take<GramWindow>(1);
    uint32_t *d_add = cv.take<uint32_t>(1);
    GramWindow w;
    map_range(m, site_begin, site_end, &w.site_begin, &w.site_end);
    const uint64_t W = window_W(m, site_begin, site_end);  // the window's ORIGINAL length
    HIP_TRY(hipMemcpyAsync(d_w, &w, sizeof w, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_W, &W, 8, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_gram_any(ctx, m, d_w, &w, 1, d_g, d_t, w.site_end - w.site_begin);
    if (rc) return rc;
    rc = launch_gram_unflip(ctx, m, d_g, 1);  // exported counts / identities: the original polarity
    if (rc) return rc;
    SimBatch b{};
    b.gram = d_g; b.stride = (uint64_t)ld * ld; b.ld = ld; b.n = n; b.W = d_W; b.kind = identity_kind; b.round_digits = -1;
    const GramWindow ow{site_begin, site_end};
    const uint32_t add_w = compact_weighted(m) ? ones_weight(m, site_begin, site_end) : 0u;
    if (m->compact) {
        if (compact_weighted(m)) {
            HIP_TRY(hipMemcpyAsync(d_add, &add_w, 4, hipMemcpyHostToDevice, ctx->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(d_ow, &ow, sizeof ow, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(seg_count_kernel, dim3(1), dim3(256), 0, ctx->stream, m->d_onesmap, d_ow, 1, (impop_window_stats *)nullptr, d_add);
            HIP_TRY(hipGetLastError());
        }
        b.add = d_add;
    }
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
    REQUIRE(params->identity_kind == IMPOP_IDENTITY_MATCH || params->identity_kind == IMPOP_IDENTITY_DICE,
            "impop_pairwise_scan: unknown identity kind");
    REQUIRE(params->round_digits <= 19, "impop_pairwise_scan: round_digits > 19 unsupported");
    REQUIRE(params->d_pi_mode >= 0 && params->d_pi_mode <= 2 && params->s_scope >= 0 && params->s_scope <= 2,
            "impop_pairwise_scan: bad d_pi_mode / s_scope");
    REQUIRE(params->fst_method <= 1, "impop_pairwise_scan: fst_method must be 0 (direct) or 1 (grouped)");
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
    // IMPOP_TRACE=1: host-side phase times of this call on stderr (where a call's time goes when the kernels are short)
    static const bool trace = [] { const char *e = getenv("IMPOP_TRACE"); return e && e[0] == '1'; }();
    const auto t_enter = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[impop_pairwise_scan] %-22s +%.1f us\n", what,
                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count());
    };
    // s_scope 2: the caller does not need S / Tajima's D (pica2- or Fst-only output): skip the site scan
    const bool want_s = params->s_scope != 2;
    if (!want_s) sp.s_scope = 0;
    // without a subset mask S comes from the matrix's cached site bitmap (s_p = s_all); with one, s_p needs the
    // subset's own counts: the streaming scan of the same windows
    const bool use_segmap = want_s && !mask_p;
    impop_scan_plan *plan = nullptr;
    int rc = (want_s && !use_segmap) ? impop_scan_plan_create(ctx, m, windows, n_windows, mask_p, mask_a, mask_b, &sp, &plan) : IMPOP_OK;
    if (rc) return rc;
    if (use_segmap) {
        rc = ensure_segmap(ctx, m);
        if (rc) return rc;
    }
    auto fail = [&](int code) {
        if (plan) impop_scan_plan_destroy(plan);
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
    std::vector<uint32_t> ia, ib;  // hud.py grouped: members of A / B with the overlap removed from both
    for (uint32_t i = 0; i < n && params->fst_method == 1; ++i) {
        if (fa[i] && !fb[i]) ia.push_back(i);
        if (fb[i] && !fa[i]) ib.push_back(i);
    }
    // ---- Gram cells.  I_ij is additive over disjoint site ranges, so overlapping (sliding) windows share the
    // Gram matrices of the elementary segments between the sorted window boundaries: every site is
    // contracted once however many windows cover it, and a window is the sum of its consecutive segments
    // (formed on the fly by the statistics kernels, SimBatch.seg_*).  Without overlap the cells are the
    // windows themselves.
    // compacted matrix: the contraction runs over the KEPT (variable) sites of each window; the dropped all-ones
    // sites come back as a per-window constant (SimBatch.add), the dropped all-zero sites contribute nothing
    lap("masks");
    std::vector<impop_window> mw;
    rc = map_windows_device(ctx, m, windows, n_windows, mw);
    if (rc) return fail(rc);
    lap("map_windows");
    struct Cell { uint64_t b, e; };
    std::vector<Cell> cells;                                // all Gram cells, in site order when segmented
    std::vector<uint32_t> first(n_windows, 0), count(n_windows, 0);
    std::vector<uint64_t> ord(n_windows);
    for (uint64_t i = 0; i < n_windows; ++i) ord[i] = i;
    // the usual window list — a BED tiling: sorted, no two windows overlapping — has nothing to share: its cells are the windows
    // (one O(n) check instead of the sort + searches below, 0.15 ms of host time per 4096 windows while the GPU waits)
    bool tiling = n_windows < 0xFFFFFFF0ull;
    for (uint64_t i = 1; i < n_windows && tiling; ++i) tiling = mw[i].site_begin >= mw[i - 1].site_end && mw[i].site_end >= mw[i].site_begin;
    if (tiling) {
        cells.resize(n_windows);
        for (uint64_t i = 0; i < n_windows; ++i) {
            cells[i] = {mw[i].site_begin, mw[i].site_end};
            first[i] = (uint32_t)i;
            count[i] = 1;
        }
    } else {
        std::vector<uint64_t> cuts;
        uint64_t win_sites = 0;
        for (uint64_t i = 0; i < n_windows; ++i)
            if (mw[i].site_end > mw[i].site_begin) {
                cuts.push_back(mw[i].site_begin);
                cuts.push_back(mw[i].site_end);
                win_sites += mw[i].site_end - mw[i].site_begin;
            }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        auto at = [&](uint64_t s) { return (size_t)(std::lower_bound(cuts.begin(), cuts.end(), s) - cuts.begin()); };
        std::vector<int64_t> cover(cuts.size() + 1, 0);
        for (uint64_t i = 0; i < n_windows; ++i)
            if (mw[i].site_end > mw[i].site_begin) {
                cover[at(mw[i].site_begin)] += 1;
                cover[at(mw[i].site_end)] -= 1;
            }
        std::vector<uint32_t> seg_before(cuts.size() + 1, 0);  // covered intervals left of cut k
        uint64_t seg_sites = 0;
        int64_t depth = 0;
        std::vector<Cell> segs;
        for (size_t k = 0; k + 1 < cuts.size(); ++k) {
            seg_before[k] = (uint32_t)segs.size();
            depth += cover[k];
            if (depth > 0) {
                segs.push_back({cuts[k], cuts[k + 1]});
                seg_sites += cuts[k + 1] - cuts[k];
            }
        }
        if (!cuts.empty()) seg_before[cuts.size() - 1] = (uint32_t)segs.size();
        if (seg_sites * 20 < win_sites * 19 && segs.size() < 0xFFFFFFF0ull) {  // >= 5 % of the contraction is shared
            cells.swap(segs);
            for (uint64_t i = 0; i < n_windows; ++i)
                if (mw[i].site_end > mw[i].site_begin) {
                    first[i] = seg_before[at(mw[i].site_begin)];
                    count[i] = seg_before[at(mw[i].site_end)] - first[i];
                }
            std::stable_sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) {
                const bool ea = count[a] == 0, eb = count[b] == 0;  // empty windows last
                return ea != eb ? eb : (!ea && first[a] < first[b]);
            });
        } else {
            if (n_windows >= 0xFFFFFFF0ull) {
                set_error("impop_pairwise_scan: too many windows");
                return fail(IMPOP_E_INVALID);
            }
            cells.resize(n_windows);
            for (uint64_t i = 0; i < n_windows; ++i) {
                cells[i] = {mw[i].site_begin, mw[i].site_end};
                first[i] = (uint32_t)i;
                count[i] = 1;
            }
        }
    }
    // Chunks of consecutive (in `ord`) windows whose cells fit the Gram scratch (<= ~8 GiB of 288): large
    // chunks keep the persistent Gram grid's last, partially filled round of tasks small next to the launch
    // weighted matrices need a second Gram buffer (the plane partials) only where the planes are NOT walked inside the
    // Gram task: the int8 kernel, or a window as heavy as 2^24 (launch_gram_any)
    bool plane_buffer = !m->wt_prefix.empty();
    if (plane_buffer && gram_planes_in_task()) {
        uint64_t heaviest = 0;  // a Gram cell is a window or a piece of one: never heavier
        for (uint64_t i = 0; i < n_windows; ++i) heaviest = std::max(heaviest, window_W(m, windows[i].site_begin, windows[i].site_end));
        if (heaviest < GRAM_FUSED_WEIGHT_LIMIT) plane_buffer = false;
    }
    lap("cells");
    const size_t gram_bytes = (size_t)ld * ld * 4;
    uint64_t cap = ((plane_buffer ? 4ull : 8ull) << 30) / gram_bytes;  // (a chromosome of 50 kb windows — 4854 on chr2 — is one chunk)
    if (cap > 8192) cap = 8192;
    cap = std::min<uint64_t>(cap, std::max<uint64_t>(cells.size(), n_windows));  // a short call stages (and copies) short tables
    if (cap < 1) cap = 1;
    void *d = nullptr;
    const size_t need = 4096 + cap * ((plane_buffer ? 2 : 1) * gram_bytes + sizeof(GramWindow) + 24 + sizeof(Pica2Out) + sizeof(HfstOut) +
                                      sizeof(impop_window_stats) + sizeof(impop_pairwise_stats) + 2 * sizeof(GramWindow) + 4 + 3584) + 16 * 256 +
                        (size_t)n * 16 + 8192;
    rc = ctx_scratch(ctx, need, &d);
    if (rc) return fail(rc);
    Carve2 cv(d);
    int32_t *d_g = cv.take<int32_t>(cap * (size_t)ld * ld);
    int32_t *d_gt = plane_buffer ? cv.take<int32_t>(cap * (size_t)ld * ld) : nullptr;
    // per-chunk metadata: ONE contiguous region mirrored on the host, so that a chunk costs one host-to-device copy
    // (eight small pageable copies were ~0.3 ms of host time between two Gram launches)
    auto up256 = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_w = 0, o_W = o_w + up256(cap * sizeof(GramWindow)), o_L = o_W + up256(cap * 8), o_first = o_L + up256(cap * 8),
                 o_count = o_first + up256(cap * 4), o_s = o_count + up256(cap * 4), o_sw = o_s + up256(cap * sizeof(impop_window_stats)),
                 o_ow = o_sw + up256(cap * sizeof(GramWindow)), meta_bytes = o_ow + up256(cap * sizeof(GramWindow));
    char *d_meta = cv.take<char>(meta_bytes);
    GramWindow *d_w = reinterpret_cast<GramWindow *>(d_meta + o_w);
    uint64_t *d_W = reinterpret_cast<uint64_t *>(d_meta + o_W);
    uint64_t *d_L = reinterpret_cast<uint64_t *>(d_meta + o_L);
    uint32_t *d_first = reinterpret_cast<uint32_t *>(d_meta + o_first);
    uint32_t *d_count = reinterpret_cast<uint32_t *>(d_meta + o_count);
    Pica2Out *d_p = cv.take<Pica2Out>(cap);
    HfstOut *d_h = cv.take<HfstOut>(cap);
    impop_window_stats *d_s = reinterpret_cast<impop_window_stats *>(d_meta + o_s);
    impop_pairwise_stats *d_o = cv.take<impop_pairwise_stats>(cap);
    GramWindow *d_sw = reinterpret_cast<GramWindow *>(d_meta + o_sw);  // the chunk's WINDOWS (d_w holds its Gram cells), matrix coordinates
    GramWindow *d_ow = reinterpret_cast<GramWindow *>(d_meta + o_ow);  // the same windows in ORIGINAL coordinates (compacted matrices)
    uint32_t *d_add = cv.take<uint32_t>(cap);     // compacted: dropped all-ones sites per window
    uint32_t *d_idx = cv.take<uint32_t>(n ? n : 1);
    uint8_t *d_fa = cv.take<uint8_t>(n ? n : 1);
    uint8_t *d_fb = cv.take<uint8_t>(n ? n : 1);
    uint32_t *d_ia = cv.take<uint32_t>(n ? n : 1);
    uint32_t *d_ib = cv.take<uint32_t>(n ? n : 1);
    hipError_t e;
#define PW_TRY(expr) \
    if ((e = (expr)) != hipSuccess) return fail(hip_fail(e, #expr, __FILE__, __LINE__))
    if (nP) PW_TRY(hipMemcpyAsync(d_idx, idx.data(), (size_t)nP * 4, hipMemcpyHostToDevice, ctx->stream));
    PW_TRY(hipMemcpyAsync(d_fa, fa.data(), n, hipMemcpyHostToDevice, ctx->stream));
    PW_TRY(hipMemcpyAsync(d_fb, fb.data(), n, hipMemcpyHostToDevice, ctx->stream));
    if (!ia.empty()) PW_TRY(hipMemcpyAsync(d_ia, ia.data(), ia.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!ib.empty()) PW_TRY(hipMemcpyAsync(d_ib, ib.data(), ib.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<impop_window_stats> scan_host;  // only with a scan plan; else the records are n_sites and zeros (S: the device fills it in)
    if (plan) {
        scan_host.resize(n_windows);
        rc = impop_scan_plan_launch(plan, nullptr);
        if (rc) return fail(rc);
        rc = impop_scan_plan_fetch(plan, scan_host.data());
        if (rc) return fail(rc);
    }
    // even out the chunks: a total slightly above the capacity would otherwise leave a last chunk of a few
    // windows whose single-workgroup epilogue kernels cost their full latency
    uint64_t cell_limit = cap;
    if (cells.size() > cap) {
        uint32_t widest = 1;
        for (uint64_t i = 0; i < n_windows; ++i) widest = std::max(widest, count[i]);
        uint64_t n_chunks = (cells.size() + cap - 1) / cap;
        if ((cells.size() + n_chunks - 1) / n_chunks + widest > cap) ++n_chunks;  // neighbours re-contract up to `widest` cells
        cell_limit = std::min<uint64_t>(cap, (cells.size() + n_chunks - 1) / n_chunks + widest);
    }
    lap("scratch + scan_host");
    // page-locked staging for the metadata going up and the records coming down (ctx_pinned)
    const size_t out_off = up256(meta_bytes);
    void *pin = nullptr;
    rc = ctx_pinned(ctx, out_off + cap * sizeof(impop_pairwise_stats), &pin);
    if (rc) return fail(rc);
    char *hmeta = reinterpret_cast<char *>(pin);
    memset(hmeta, 0, meta_bytes);
    GramWindow *gw = reinterpret_cast<GramWindow *>(hmeta + o_w), *swv = reinterpret_cast<GramWindow *>(hmeta + o_sw),
               *owv = reinterpret_cast<GramWindow *>(hmeta + o_ow);
    uint64_t *Wv = reinterpret_cast<uint64_t *>(hmeta + o_W), *Lv = reinterpret_cast<uint64_t *>(hmeta + o_L);
    uint32_t *fv = reinterpret_cast<uint32_t *>(hmeta + o_first), *cvv = reinterpret_cast<uint32_t *>(hmeta + o_count);
    impop_window_stats *sv = reinterpret_cast<impop_window_stats *>(hmeta + o_s);
    impop_pairwise_stats *ov = reinterpret_cast<impop_pairwise_stats *>(hmeta + out_off);
    std::vector<uint32_t> add_h;
    uint64_t call_max_W = 0;  // bounds every Gram count of the call (a cell is a window or a piece of one; compacted: + its constant)
    for (uint64_t i = 0; i < n_windows; ++i) call_max_W = std::max(call_max_W, window_W(m, windows[i].site_begin, windows[i].site_end));
    bool g16 = false;
    for (uint64_t base = 0; base < n_windows;) {
        // windows ord[base .. base+cnt): their cells are [c_lo, c_hi)
        uint64_t cnt = 0;
        uint32_t c_lo = 0, c_hi = 0;
        bool have = false;
        while (base + cnt < n_windows && cnt < cell_limit) {
            const uint64_t wdx = ord[base + cnt];
            if (count[wdx]) {
                const uint32_t lo = have ? std::min(c_lo, first[wdx]) : first[wdx];
                const uint32_t hi = have ? std::max(c_hi, first[wdx] + count[wdx]) : first[wdx] + count[wdx];
                if ((uint64_t)(hi - lo) > (cnt == 0 ? cap : cell_limit)) {
                    if (cnt == 0) {
                        set_error("impop_pairwise_scan: window %llu spans %u segments, more than the %llu Gram matrices that fit "
                                  "the scratch", (unsigned long long)wdx, count[wdx], (unsigned long long)cap);
                        return fail(IMPOP_E_INVALID);
                    }
                    break;
                }
                c_lo = lo; c_hi = hi; have = true;
            }
            ++cnt;
        }
        const uint32_t n_cells = have ? c_hi - c_lo : 0;
        uint64_t max_sites = 0;
        for (uint32_t c = 0; c < n_cells; ++c) {
            gw[c] = {cells[c_lo + c].b, cells[c_lo + c].e};
            max_sites = std::max<uint64_t>(max_sites, cells[c_lo + c].e - cells[c_lo + c].b);
        }
        // the Gram launch needs the cells alone: they go up first and the kernel starts, the per-window tables are filled in (and
        // copied) while it runs
        if (n_cells) PW_TRY(hipMemcpyAsync(d_meta + o_w, hmeta + o_w, (size_t)n_cells * sizeof(GramWindow), hipMemcpyHostToDevice, ctx->stream));
        if (n_cells) {
            hipEvent_t ev1 = nullptr;
            if (ctx->gram_timing) {  // impop_ctx_gram_timing: the Gram launch(es) of this chunk between two events
                if (ctx->gram_events_used == ctx->gram_events.size()) {
                    hipEvent_t a, b;
                    PW_TRY(hipEventCreate(&a));
                    PW_TRY(hipEventCreate(&b));
                    ctx->gram_events.push_back({a, b});
                }
                PW_TRY(hipEventRecord(ctx->gram_events[ctx->gram_events_used].first, ctx->stream));
                ev1 = ctx->gram_events[ctx->gram_events_used].second;
                ctx->gram_events_used++;
            }
            // counts as uint16 where every count of the call fits (a count is at most its window's W): half the result bytes
            static const bool u16_off = [] { const char *e = getenv("IMPOP_GRAM_U16"); return e && e[0] == '0'; }();
            g16 = !u16_off && call_max_W < 65536;
            rc = launch_gram_any(ctx, m, d_w, gw, n_cells, d_g, d_gt, max_sites, &g16);
            if (rc) return fail(rc);
            if (ev1) PW_TRY(hipEventRecord(ev1, ctx->stream));
            if (params->identity_kind != IMPOP_IDENTITY_MATCH) {  // `match` sees Hamming distances only: polarity-invariant
                rc = launch_gram_unflip(ctx, m, d_g, n_cells, g16);
                if (rc) return fail(rc);
            }
        }
        for (uint64_t k = 0; k < cnt; ++k) {
            const uint64_t wdx = ord[base + k];
            Wv[k] = window_W(m, windows[wdx].site_begin, windows[wdx].site_end);
            Lv[k] = windows[wdx].seq_len;
            fv[k] = count[wdx] ? first[wdx] - c_lo : 0;
            cvv[k] = count[wdx];
            if (plan) sv[k] = scan_host[wdx];
            else { memset(&sv[k], 0, sizeof(sv[k])); sv[k].n_sites = (uint32_t)Wv[k]; }
            swv[k] = {mw[wdx].site_begin, mw[wdx].site_end};
            owv[k] = {windows[wdx].site_begin, windows[wdx].site_end};
        }
        // problem k IS Gram matrix k (disjoint windows, none empty): the epilogue kernels then take their one-matrix variants
        bool one_to_one = true;
        for (uint64_t k = 0; k < cnt && one_to_one; ++k) one_to_one = cvv[k] == 1 && fv[k] == k;
        lap("chunk metadata");
        PW_TRY(hipMemcpyAsync(d_meta + o_W, hmeta + o_W, meta_bytes - o_W, hipMemcpyHostToDevice, ctx->stream));
        if (use_segmap) {
            hipLaunchKernelGGL(seg_count_kernel, dim3((uint32_t)((cnt + 3) / 4)), dim3(256), 0, ctx->stream, m->d_segmap, d_sw, cnt, d_s,
                               (uint32_t *)nullptr);
            PW_TRY(hipGetLastError());
        }
        SimBatch b{};
        b.gram = d_g; b.stride = (uint64_t)ld * ld; b.ld = ld; b.n = n; b.W = d_W; b.kind = params->identity_kind;
        b.g16 = g16 ? 1u : 0u;
        b.max_W = call_max_W;
        b.round_digits = params->round_digits < 0 ? -1 : params->round_digits;
        b.seg_first = one_to_one ? nullptr : d_first; b.seg_count = one_to_one ? nullptr : d_count;
        if (compact_weighted(m)) {  // the dropped all-ones sites' summed weights, from the host prefix sums
            add_h.resize(cnt);
            for (uint64_t k = 0; k < cnt; ++k) add_h[k] = ones_weight(m, windows[ord[base + k]].site_begin, windows[ord[base + k]].site_end);
            PW_TRY(hipMemcpyAsync(d_add, add_h.data(), cnt * 4, hipMemcpyHostToDevice, ctx->stream));
            b.add = d_add;
        } else if (m->compact) {    // ... their count, from the bitmap on the device
            hipLaunchKernelGGL(seg_count_kernel, dim3((uint32_t)((cnt + 3) / 4)), dim3(256), 0, ctx->stream, m->d_onesmap, d_ow, cnt,
                               (impop_window_stats *)nullptr, d_add);
            PW_TRY(hipGetLastError());
            b.add = d_add;
        }
        // pica2 grouping and the Fst sums are independent, latency-bound one-workgroup-per-window kernels: pica2 goes
        // to the side stream (fork behind the Gram launch, join before the finalize) so the two overlap
        static const bool use_side = [] { const char *e = getenv("IMPOP_PW_SIDE_STREAM"); return !(e && e[0] == '0'); }();
        if (!use_side) {  // A/B switch (tools/): the two kernels one after the other on the main stream
            rc = launch_pica2(ctx, b, cnt, mask_p ? d_idx : nullptr, nP, nullptr, params->threshold, d_L, d_p, nullptr);
            if (rc) return fail(rc);
        } else {
            if (!ctx->side) {
                PW_TRY(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
                PW_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                PW_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            PW_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
            PW_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->side;
            rc = launch_pica2(ctx, b, cnt, mask_p ? d_idx : nullptr, nP, nullptr, params->threshold, d_L, d_p, nullptr);
            ctx->stream = main_stream;
            if (rc) return fail(rc);
            PW_TRY(hipEventRecord(ctx->ev_join, ctx->side));
        }
        if (params->fst_method == 1)
            rc = launch_hud_grouped(ctx, b, cnt, d_ia, (uint32_t)ia.size(), d_ib, (uint32_t)ib.size(), nullptr, nullptr, params->threshold,
                                        d_L, d_h);
        else
            rc = launch_hfst(ctx, b, cnt, d_fa, d_fb, d_L, d_h);
        if (rc) return fail(rc);
        if (use_side) PW_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        PairFinalIn in{d_p, d_h, d_s};
        rc = ensure_tajima_consts(ctx, nP >= 2 ? (int64_t)nP : 2);  // the cache may have been retargeted by another plan
        if (rc) return fail(rc);
        hipLaunchKernelGGL(pairwise_finalize_kernel, dim3((uint32_t)((cnt + 63) / 64)), dim3(64), 0, ctx->stream, in, cnt,
                           want_s ? nP : 0u, params->d_pi_mode, want_s ? params->s_scope : 0, ctx->d_taj, d_o);
        PW_TRY(hipGetLastError());
        lap("chunk launched");
        PW_TRY(hipMemcpyAsync(ov, d_o, cnt * sizeof(impop_pairwise_stats), hipMemcpyDeviceToHost, ctx->stream));
        rc = ctx_err_fetch(ctx);
        if (rc) return fail(rc);
        PW_TRY(hipStreamSynchronize(ctx->stream));  // the staging vectors are reused by the next chunk
        rc = ctx_err_result(ctx, "impop_pairwise_scan");  // a device-side consistency check tripped: no partial results
        if (rc) return fail(rc);
        for (uint64_t k = 0; k < cnt; ++k) out_host[ord[base + k]] = ov[k];
        base += cnt;
        lap("chunk done");
    }
#undef PW_TRY
    if (plan) impop_scan_plan_destroy(plan);
    return IMPOP_OK;
}
