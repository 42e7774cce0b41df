// internal.h — shared host-side structures of libimpop_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/impop_hip.h"

#define IMPOP_API extern "C" __attribute__((visibility("default")))

namespace impop {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return impop::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define REQUIRE(cond, ...)                 \
    do {                                   \
        if (!(cond)) {                     \
            impop::set_error(__VA_ARGS__); \
            return IMPOP_E_INVALID;        \
        }                                  \
    } while (0)

// per-site outputs and the all-pairs path need every site: they refuse compacted matrices
#define NOT_COMPACT(m, fn)                                                                                     \
    do {                                                                                                       \
        if ((m)->compact) {                                                                                    \
            impop::set_error("%s: not available on a compacted matrix (impop_matrix_compact drops monomorphic sites)", fn); \
            return IMPOP_E_UNSUPPORTED;                                                                        \
        }                                                                                                      \
    } while (0)

void matrix_drop_derived(const impop_matrix *m);  // frees the lazily built weight planes / masked operand / site bitmap

#define NOT_WEIGHTED(m, fn)                                                                  \
    do {                                                                                     \
        if (!(m)->wt_prefix.empty()) {                                                                   \
            impop::set_error("%s: not available on a matrix with site weights", fn);         \
            return IMPOP_E_UNSUPPORTED;                                                      \
        }                                                                                    \
    } while (0)

// ---- SB64: site-blocked, wave-interleaved layout --------------------------------
// The site axis is cut into blocks of 64 sites (one wavefront).  A site holds
// wps = ceil(n_hap/32) dwords (dword k = haplotypes 32k..32k+31).  Inside a block the
// dwords are stored in 16-byte granules interleaved over the 64 sites, so that lane l
// of a wave (= site 64b+l) reads its granule g with ONE fully coalesced 1 KiB
// global_load_dwordx4:
//     dword(b, l, k) @ b*64*wps + (k/4)*256 + l*4 + (k%4)          for k/4 < G-1
//     dword(b, l, k) @ b*64*wps + (G-1)*256 + l*r + (k - 4(G-1))    last granule, r dwords
// with G = ceil(wps/4), r = wps - 4(G-1) in 1..4.  Bytes per site = 4*wps exactly
// (60 B for 465 haplotypes: 3.2 % above the algorithmic n/8).
struct SbGeom {
    uint32_t n_hap = 0;
    uint32_t wps = 0;    // dwords per site
    uint32_t G = 0;      // granules per site
    uint32_t r = 0;      // dwords in the last granule
    uint64_t n_site = 0;
    uint64_t n_block = 0;  // ceil(n_site/64)
};

// dword index of (block b, lane/site l, dword k) in the SB64 layout
__host__ __device__ inline uint64_t sb_index(uint32_t wps, uint32_t G, uint32_t r, uint64_t b, uint32_t l,
                                             uint32_t k) {
    const uint32_t g = k >> 2;
    const uint64_t base = b * 64ull * wps;
    return (g + 1 < G) ? base + (uint64_t)g * 256 + l * 4 + (k & 3)
                       : base + (uint64_t)(G - 1) * 256 + (uint64_t)l * r + (k - 4 * (G - 1));
}

}  // namespace impop

struct impop_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char arch[128] = {0};
    int n_cu = 0;
    // cached Tajima constants on the device, keyed by n (8 doubles: a1,a2,b1,b2,c1,c2,e1,e2)
    double *d_taj = nullptr;
    int64_t taj_n = -1;
    uint32_t *d_queue = nullptr;  // 8 task-queue heads of the persistent Gram kernel
    // device error word (context.hip: ctx_err_word / ctx_err_fetch / ctx_err_result): kernels OR a bit in when an internal
    // invariant fails (stats.hip: the grouping's progress bound), the call that launched them returns IMPOP_E_INTERNAL
    uint32_t *d_err = nullptr;
    uint32_t h_err = 0;
    // impop_ctx_gram_timing: event pairs around the Gram launches of impop_pairwise_scan
    bool gram_timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gram_events;
    size_t gram_events_used = 0;
    // side stream + fork/join events (created on first use): independent latency-bound epilogue kernels of the
    // all-pairs path run next to each other instead of one after the other
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // growable scratch
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // growable PINNED host staging (context.hip ctx_pinned): per-chunk metadata up and records down of impop_pairwise_scan —
    // from pageable memory either copy is a host memcpy into the runtime's own staging buffer first (0.8 MB up before the Gram
    // kernel can start, 0.4 MB down after the last kernel, per 4096 windows)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // growable side buffers of the epilogue kernels that split large problems over several workgroups (stats.hip):
    // slot 0 h-fst partial sums, slot 1 pica2 group tables + row sums; separate because the two run side by side
    void *d_aux[2] = {nullptr, nullptr};
    size_t aux_bytes[2] = {0, 0};
};

struct impop_matrix {
    impop::SbGeom g;
    uint32_t *d_sb = nullptr;   // SB64 layout, n_block*64*wps dwords
    uint64_t sb_bytes = 0;
    // RB32 — row-group-blocked hap-major copy, operand layout of the Gram kernel (optional,
    // IMPOP_KEEP_HAP_MAJOR): rows are grouped by 32, sites by 64-site cells (2 dwords), and one
    // (group, cell) holds the 32 rows' dword pairs contiguously (256 B = one coalesced wave load):
    //     dword(row, d) @ (((row>>5) * rb_nb + (d>>1)) * 32 + (row&31)) * 2 + (d&1)
    uint32_t *d_rb = nullptr;
    uint64_t rb_nb = 0;         // cells per row group incl. 4 cells of slack (prefetch)
    uint32_t n_hap_pad = 0;     // rows padded to a multiple of 96 (zero rows; Gram tiles are 96 wide)
    // RB32 in minor-allele polarity (layout.hip sb_to_hm_kernel): row phi_row (= n_hap, inside the padding) holds the set of
    // complemented sites; 0xFFFFFFFF = stored as given (no padding row free: n_hap a multiple of 96; or IMPOP_NO_POLARITY=1)
    uint32_t phi_row = 0xFFFFFFFFu;
    uint64_t rb_bytes = 0;
    // compacted matrix (impop_matrix_compact): only the sites variable among all haplotypes were kept;
    // pos[k] = original index of kept site k (host copy for window mapping), n_site_orig = original length
    uint32_t *d_wt = nullptr;  // optional per-site weights (impop_matrix_set_site_weights), plain site order
    // weighted all-pairs path (pairwise.hip), built on first use: bit planes of the weights over 32-site dwords
    // (plane k, dword d: bit j = bit k of weight[32 d + j]) and one masked copy of the RB32 operand
    mutable uint32_t *d_wplanes = nullptr;
    mutable uint32_t wplane_bits = 0;     // planes that have any bit set
    mutable uint64_t wplane_stride = 0;   // dwords per plane
    mutable uint32_t *d_rb_masked = nullptr;
    // lazily built bitmap of the sites that segregate among ALL haplotypes (bit s of dword s>>5), cached for the
    // all-pairs path's S (pairwise.hip); dropped with the matrix
    mutable uint32_t *d_segmap = nullptr;
    // compacted matrix built from one that kept its hap-major copy: bitmap, in ORIGINAL site coordinates, of the
    // dropped sites that EVERY haplotype carries (c_s = n): each adds 1 to every I_ij, so the all-pairs path on the
    // variable sites alone plus this per-window count is exact (pairwise.hip)
    uint32_t *d_onesmap = nullptr;
    uint64_t *d_pos = nullptr;  // compacted: device copy of `pos` (map_windows_device: window edges -> kept-site indices on the GPU)
    bool compact = false;
    uint64_t n_site_orig = 0;
    std::vector<uint64_t> pos;
    std::vector<uint64_t> pos_coarse;  // pos[k << POS_COARSE_SHIFT]: a cache-resident first level for pos_lower_bound
    // compacted from a WEIGHTED matrix that kept its hap-major copy (all-pairs path): prefix sums, in ORIGINAL coordinates, of
    // the weights of the dropped sites every haplotype carries (a window's constant `add`), and of the kept columns' weights
    std::vector<uint64_t> ones_wt_prefix, kept_wt_prefix;
    std::vector<uint64_t> wt_prefix;  // weighted: prefix sums of the (ORIGINAL, if compacted) site weights, n + 1 entries
    int device = 0;
    mutable int users = 0;      // live scan plans referencing this matrix (impop_matrix_free refuses while > 0)
};

namespace impop {
int ctx_scratch(impop_ctx *ctx, size_t bytes, void **out);
int ctx_pinned(impop_ctx *ctx, size_t bytes, void **out);  // host, page-locked, grow-only; valid until the next larger request
int ctx_err_fetch(impop_ctx *ctx);                 // enqueue its copy to the host (before the call's own stream sync)
int ctx_err_result(impop_ctx *ctx, const char *fn);  // after that sync: IMPOP_OK, or IMPOP_E_INTERNAL (word cleared, message set)
constexpr uint32_t DEV_ERR_GROUPING = 1u;            // greedy_groups_bits ran out of its progress bound
int ctx_aux(impop_ctx *ctx, int slot, size_t bytes, void **out);
// pairwise.hip: build (once) the bitmap of the sites that segregate among all haplotypes, m->d_segmap
int ensure_segmap(impop_ctx *ctx, const impop_matrix *m);
int ensure_tajima_consts(impop_ctx *ctx, int64_t n);  // fills ctx->d_taj for n (device kernel)

// windows are given in ORIGINAL site coordinates; for a compacted matrix map them to kept-site index
// ranges (`mapped`), else `mapped` is a plain copy.  span() = the coordinate range windows must lie in.
inline uint64_t matrix_span(const impop_matrix *m) { return m->compact ? m->n_site_orig : m->g.n_site; }
void map_windows(const impop_matrix *m, const impop_window *windows, uint64_t n, std::vector<impop_window> &mapped);
// the same on the device (one thread per window edge, binary search in the device copy of the positions): 8192 searches over
// 87 MB of positions cost the host 0.55 ms per call — as long as the Gram launch they precede — and the GPU some 50 us
int map_windows_device(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n, std::vector<impop_window> &mapped);
// compacted matrices: index of the first kept site at or right of original coordinate s (= number of kept sites left of s)
uint64_t pos_lower_bound(const impop_matrix *m, uint64_t s);
constexpr unsigned POS_COARSE_SHIFT = 12;

// layout.hip
int launch_hm_to_sb(impop_ctx *ctx, const uint32_t *d_hm, uint64_t hm_stride, const SbGeom &g, uint32_t *d_sb);
// rb_nb == 0: plain hap-major rows of hm_stride dwords; else RB32 addressing with rb_nb cells per row group
int launch_sb_to_hm(impop_ctx *ctx, const uint32_t *d_sb, const SbGeom &g, uint64_t blk_begin, uint64_t blk_end,
                    uint32_t *d_hm, uint64_t hm_stride, uint32_t n_rows, uint64_t rb_nb = 0, uint32_t phi_row = 0xFFFFFFFFu);

}  // namespace impop
