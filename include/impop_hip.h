/*
 * impop_hip.h — C ABI of libimpop_hip.so: the MI355X (gfx950) windowed
 * population-statistics engine for impop's pairwise-diversity hot path.
 *
 * The reference (pangenome/impop) has no FFI: the path sits behind Python
 * functions and CLIs.  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).
 * The reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md; impop_amd/_lib.py is exactly that stub.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no exceptions, no Python or
 *    torch types cross the boundary.
 *  - every function returns 0 on success or a negative impop_status; the
 *    message is retrievable with impop_last_error() (thread-local).
 *  - a context (impop_ctx) owns one HIP stream on one device; it is not
 *    thread-safe, the library is re-entrant across contexts.
 *  - host pointers are caller-owned; device memory lives behind opaque handles.
 *  - haplotypes are indexed 0..n-1 in the caller's order; for the name-ordered
 *    semantics of pica2.py/af.py the caller passes them in lexicographic name
 *    order (the Python layer does).
 *  - dense identity matrices are row-major double[n*n]; NaN marks a pair that
 *    is absent from the .sim table (pica2.py:131-134, h-fst.py:152-153).
 *  - all compute runs on the GPU; there is no CPU fallback.  Without a usable
 *    gfx950 device impop_ctx_create fails with IMPOP_E_NODEVICE.
 */
#ifndef IMPOP_HIP_H
#define IMPOP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMPOP_ABI_VERSION 2

typedef enum impop_status {
    IMPOP_OK = 0,
    IMPOP_E_INVALID = -1,   /* bad argument (also the reference's ValueError cases, tj_d.py:48-51) */
    IMPOP_E_NODEVICE = -2,  /* no HIP device / wrong architecture */
    IMPOP_E_HIP = -3,       /* a HIP runtime call failed */
    IMPOP_E_NOMEM = -4,
    IMPOP_E_UNSUPPORTED = -5,
    IMPOP_E_INTERNAL = -6   /* a device-side consistency check tripped (results of the call are not to be used) */
} impop_status;

typedef struct impop_ctx impop_ctx;
typedef struct impop_matrix impop_matrix;
typedef struct impop_scan_plan impop_scan_plan;

/* ---- library / context -------------------------------------------------- */
int impop_version(void);                 /* IMPOP_ABI_VERSION */
const char *impop_last_error(void);      /* thread-local, never NULL */
int impop_device_count(int *count);
/* stream: an existing hipStream_t created by the caller (its ordering is then the caller's: collectives
 * and copies issued on it see the scans), or NULL to create a private non-blocking stream.  NULL is also
 * HIP's legacy null stream — it cannot be adopted; a caller that works on the null stream (torch's default
 * stream reports handle 0) must create an explicit stream and pass that. */
int impop_ctx_create(int device, void *stream, impop_ctx **out);
int impop_ctx_destroy(impop_ctx *ctx);
int impop_ctx_synchronize(impop_ctx *ctx);
/* device name/arch string of the context's device, e.g. "gfx950:sramecc+:xnack-" */
int impop_ctx_device_name(impop_ctx *ctx, char *buf, size_t buflen);

/* ---- presence matrix ------------------------------------------------------
 * Replaces the per-window `impg similarity` -> .sim TSV -> read_similarity_file
 * round trip (run_pica2_impg.sh:162-175, pica2.py:6-58, h-fst.py:84-119): the
 * haplotype x site presence matrix stays resident in HBM.
 *
 * Interchange layout ("hap-major"): bits[i*row_stride_words + (s>>6)] bit (s&63)
 * is 1 iff haplotype i carries the allele at site s.
 * keep flags: which device layouts to materialise. */
#define IMPOP_KEEP_SITE_BLOCKED 1u /* SB64 layout used by impop_scan (always kept) */
#define IMPOP_KEEP_HAP_MAJOR 2u    /* haplotype-major (row-group-blocked) copy needed by impop_pairwise_* */

int impop_matrix_upload(impop_ctx *ctx, const uint64_t *bits_hap_major, uint32_t n_hap, uint64_t n_site,
                        uint64_t row_stride_words, uint32_t keep_flags, impop_matrix **out);

/* Synthetic matrix generated on the device (bench / tests; SURVEY.md §8d
 * generator re-expressed as a counter-based hash so any window can be
 * regenerated independently on the CPU): ancestral allele per site, n_founder
 * lineages each differing at p_founder of sites, per-32-haplotype-word private
 * flips with probability p_private_word. */
typedef struct impop_synth_params {
    uint64_t seed;
    uint32_t n_founder;       /* 1..32 */
    double p_founder;         /* per founder per site */
    double p_private_word;    /* per (site, 32-haplotype word): one random bit flips */
} impop_synth_params;
int impop_matrix_synthetic(impop_ctx *ctx, uint32_t n_hap, uint64_t n_site, const impop_synth_params *p,
                           uint32_t keep_flags, impop_matrix **out);
/* Sites [site_begin, site_begin + n_site) of the same synthetic chromosome (the generator is counter-based on the global
 * site index): what a rank of a sharded scan holds — its window range's slab plus halo — without anybody
 * ever materialising the whole matrix.  Site 0 of the returned matrix is global site `site_begin`. */
int impop_matrix_synthetic_slab(impop_ctx *ctx, uint32_t n_hap, uint64_t site_begin, uint64_t n_site,
                                const impop_synth_params *p, uint32_t keep_flags, impop_matrix **out);

/* Copy sites [site_begin, site_end) back to the host in hap-major layout
 * (bit 0 of word 0 of each row = site_begin). */
int impop_matrix_download(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                          uint64_t *bits_hap_major_out, uint64_t row_stride_words);
/* Optional per-site weights (NULL removes them): column s stands for weights[s] base pairs, e.g. one
 * column per graph node with the node's length, instead of repeating the column per bp.  impop_scan /
 * impop_scan_plan_* then return sum_s w_s c(n-c) sums and n_sites = sum_s w_s over the window — the
 * records of the bp-expanded matrix — while s_all / s_p / s_a / s_b keep counting COLUMNS (variable nodes,
 * what a VCF of the window lists, run_tajd.sh:148).  impop_matrix_compact keeps the weights (a window's W stays
 * the sum over all its original columns); impop_scan_multi honours them too.  Set them before building plans
 * (refused while plans of this matrix are alive).  The all-pairs path (impop_pairwise_*) honours them as well:
 * I_ij = sum_s w_s b_is b_js — the bp-weighted node-sharing counts behind `impg similarity`'s identity
 * (run_pica2_impg.sh:162-175) — W = sum_s w_s, computed exactly through the weights' bit planes
 * (I = sum_k 2^k Gram(M & W_k)); a window's summed weights must stay below 2^31. */
int impop_matrix_set_site_weights(impop_ctx *ctx, impop_matrix *m, const uint32_t *weights_host);

/* Keep only the sites that are variable among ALL haplotypes (0 < c_s < n), with their original
 * positions.  Monomorphic sites add 0 to every sum_s c(n-c) of every subset and are never segregating
 * (what `povu gfa2vcf | wc -l` counts, run_tajd.sh:148), so impop_scan / impop_scan_plan_* /
 * impop_scan_multi on the compacted matrix, given windows in the ORIGINAL site coordinates, return
 * records identical to those of the full matrix (n_sites = the window's original length) while
 * streaming only the variable sites.  When `m` kept its hap-major copy (IMPOP_KEEP_HAP_MAJOR), the compacted matrix
 * serves the all-pairs path too: a dropped site that NO haplotype carries adds nothing to any I_ij, one that EVERY
 * haplotype carries adds exactly 1 — its weight w_s on a weighted matrix — to every I_ij (diagonal included), so
 * impop_pairwise_* contract the kept sites of a window only and add the window's count of dropped all-ones sites (a
 * bitmap in original coordinates; for a weighted source, host prefix sums of their weights) — identical counts,
 * identities and records from W/S times fewer multiply-adds, and on node-level matrices without the long shared
 * anchors' weight planes.
 * Per-site outputs (impop_afs, impop_site_counts, impop_ehh) return IMPOP_E_UNSUPPORTED on a compacted matrix. */
int impop_matrix_compact(impop_ctx *ctx, const impop_matrix *m, impop_matrix **out);
/* original site index of kept sites [first, first+count) of a compacted matrix; n_site_orig (nullable)
 * receives the original number of sites */
int impop_matrix_positions(const impop_matrix *m, uint64_t first, uint64_t count, uint64_t *positions_out,
                           uint64_t *n_site_orig);
int impop_matrix_info(const impop_matrix *m, uint32_t *n_hap, uint64_t *n_site, uint64_t *device_bytes,
                      uint32_t *bytes_per_site);
int impop_matrix_free(impop_ctx *ctx, impop_matrix *m);

/* ---- windowed scan: pi + Hudson Fst + Tajima's D + S in one pass ------------
 * Replaces, per window, the chain run_pica2_impg.sh:175 / run_h-fst.sh:74 /
 * run_tajd.sh:148,166,180, i.e. pica2.analyze_similarity_matrix (pica2.py:60)
 * at threshold >= 1 (every haplotype its own group), h-fst.calculate_fst
 * (h-fst.py:173) and tj_d.tajimas_d (tj_d.py:47) on the `match` identity
 * sim_ij = (W - H_ij)/W of the window's W sites, using the exact identities
 *   sum_{i<j in P} H_ij = sum_s c_P,s (n_P - c_P,s)
 *   sum_{i in A, j in B} H_ij = sum_s [c_A,s (n_B - c_B,s) + c_B,s (n_A - c_A,s)]
 * (SURVEY.md Appendix A.1), so one streaming pass over the bit matrix suffices. */
typedef struct impop_window {
    uint64_t site_begin;  /* first site of the window */
    uint64_t site_end;    /* one past the last site */
    uint64_t seq_len;     /* `-l` of pica2.py:177 / h-fst.py:278; 0 = not given */
} impop_window;

typedef struct impop_window_stats { /* 128 bytes, fixed layout */
    uint32_t n_sites;     /* W */
    uint32_t s_all;       /* #{s: 0 < c_s < n} over all rows (run_tajd.sh:126,148: un-subset graph) */
    uint32_t s_p;         /* segregating within subset P */
    uint32_t s_a, s_b;    /* segregating within A, within B */
    uint32_t flags;       /* reserved, 0 */
    uint64_t sum_p;       /* sum_s cP (nP - cP)      = sum_{i<j in P} H_ij */
    uint64_t sum_a;       /* sum_s cA (nA - cA) */
    uint64_t sum_b;       /* sum_s cB (nB - cB) */
    uint64_t sum_ab;      /* sum_s cA(nB-cB) + cB(nA-cA) = sum_{A x B} H_ij */
    double pi;            /* pica2.py:154  (mean over pairs in P of 1 - sim) */
    double pi_site;       /* pica2.py:164  pi / seq_len; NaN when seq_len == 0 (None) */
    double pi_a, pi_b, pi_xy, dxy, da; /* h-fst.py:233-249 (divided by seq_len when > 0) */
    double fst;           /* h-fst.py:214-221 */
    double tajima_d;      /* tj_d.py:47-69; NaN where the reference prints nan/NA */
} impop_window_stats;

typedef struct impop_scan_params {
    uint32_t struct_size; /* sizeof(impop_scan_params) */
    /* which pi feeds Tajima's D: 0 = as wired by run_tajd.sh:166-180 (per-site pi
     * through the "%.8f" text round trip), 1 = per-site pi unrounded,
     * 2 = mean pairwise differences pi*W (textbook; not what the reference does) */
    int32_t d_pi_mode;
    /* S used for D: 0 = all rows of the matrix (run_tajd.sh:126,148), 1 = within P */
    int32_t s_scope;
    uint32_t tile_blocks; /* 0 = default; 64-site blocks per work tile (tuning) */
} impop_scan_params;

/* Masks are n_hap-bit little-endian bitsets (uint64 words).  mask_p: the
 * sample subset for pi / D (run_tajd.sh -l list; NULL = all haplotypes);
 * mask_a / mask_b: populations for Hudson Fst (NULL = empty).  Haplotypes in
 * both A and B are removed from both (h-fst.py:181-185). */
int impop_scan_plan_create(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                           const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b,
                           const impop_scan_params *params, impop_scan_plan **out);
/* Replace the three subset masks of an existing plan (the tile tables depend on the windows only), e.g.
 * to scan the same windows for many population pairs; takes effect for launches issued afterwards. */
int impop_scan_plan_set_masks(impop_scan_plan *plan, const uint64_t *mask_p, const uint64_t *mask_a,
                              const uint64_t *mask_b);
/* Enqueue one pass over all windows on the context's stream (no host sync, no
 * allocation: graph-capturable).  d_out: device buffer of n_windows records, or
 * NULL to use the plan's internal buffer. */
int impop_scan_plan_launch(impop_scan_plan *plan, void *d_out);
/* Synchronise and copy the plan's internal result buffer to the host. */
int impop_scan_plan_fetch(impop_scan_plan *plan, impop_window_stats *out_host);
int impop_scan_plan_info(const impop_scan_plan *plan, uint64_t *n_tiles, uint64_t *bytes_streamed);
/* Measurement aid: with timing enabled every launch brackets the streaming kernel
 * (not the tiny epilogue) with hipEvents on the context's stream; elapsed() synchronises
 * and returns the summed kernel time and the number of launches since enable/reset. */
int impop_scan_plan_timing(impop_scan_plan *plan, int enable);
int impop_scan_plan_elapsed(impop_scan_plan *plan, double *total_ms, uint64_t *launches);
int impop_scan_plan_destroy(impop_scan_plan *plan);
/* Convenience: create + launch + fetch + destroy. */
int impop_scan(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
               const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b,
               const impop_scan_params *params, impop_window_stats *out_host);

/* K disjoint populations, all K(K-1)/2 Hudson Fst pairs in ONE streaming pass: replaces the
 * panel loops of run_h_fst_panels.sh:60-71 (one run_h-fst.sh per population pair).
 * masks: n_pop bitsets of ceil(n_hap/64) uint64 words each; populations must be disjoint.
 * out: n_windows x K(K-1)/2 records, pairs ordered (0,1),(0,2),...,(1,2),...; per pair the six
 * values of h-fst.py:233-249 with population k as A and l as B. */
typedef struct impop_pair_stats {
    double fst, pi_a, pi_b, pi_xy, dxy, da;
} impop_pair_stats;
int impop_scan_multi(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                     const uint64_t *masks, uint32_t n_pop, impop_pair_stats *out_host);

/* Allele-frequency spectrum per window (scripts/wip/op-afs.py): out[w*(nP+1) + c] = number of
 * sites of window w at which exactly c haplotypes of `mask` (NULL = all; nP = its size) carry the
 * allele. */
int impop_afs(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
              const uint64_t *mask, uint32_t *out_host);

/* Per-site allele counts c_s of the haplotypes in `mask` (NULL = all) for sites
 * [site_begin, site_end): the per-site allele frequency is c_s / n. */
int impop_site_counts(impop_ctx *ctx, const impop_matrix *m, const uint64_t *mask, uint64_t site_begin,
                      uint64_t site_end, uint32_t *counts_out_host);

/* Extended haplotype homozygosity, calc_EHH of scripts/wip/ehhgfa.py:6-21: out[i] =
 * round(#{pairs of `mask` members (NULL = all) identical on window sites 0..i} / (m(m-1)/2), 3),
 * i over [site_begin, site_end).  reverse != 0 walks the window from its last site backwards
 * (calc_EHH of the column-flipped matrix, ehhgfa.py:61).  m < 2 fills 500.0 (ehhgfa.py:17-18).
 * n_members (nullable) receives m. */
int impop_ehh(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end, const uint64_t *mask,
              int reverse, double *ehh_out_host, uint32_t *n_members);

/* Device address of the plan's internal record buffer (n_windows x impop_window_stats, written by launches
 * with d_out == NULL): what a caller hands to impop_gather_records without owning any device memory itself. */
int impop_scan_plan_device_records(impop_scan_plan *plan, void **d_records);

/* ---- multi-GPU -----------------------------------------------------------------
 * Replaces the serial per-window loops of run_tajd.sh:103-196, run_h-fst.sh:155-190 and run_pica2_impg.sh:126-190
 * ACROSS GPUs: no statistic spans windows, so the window list is cut into contiguous ranges (the first
 * n % shards ranges hold one window more), each GPU keeps only the slab of sites its windows touch (sliding
 * windows: slabs overlap by the halo) and the only exchange is ONE all-gather of the 128-byte records. */

/* items [first, first + count) of shard `shard` out of n_shards */
int impop_shard_range(uint64_t n_items, int n_shards, int shard, uint64_t *first, uint64_t *count);
/* the windows of a shard and the site range [slab_begin, slab_end) they touch (all outputs nullable) */
int impop_shard_windows(const impop_window *windows, uint64_t n_windows, int n_shards, int shard, uint64_t *first_window,
                        uint64_t *n_shard_windows, uint64_t *slab_begin, uint64_t *slab_end);

/* One process driving n_ctx devices (or n_ctx contexts of one device).  slabs[k] lives on ctxs[k] and holds the
 * sites [slab_site_begin[k], slab_site_begin[k] + its n_site) of the chromosome — at least the range
 * impop_shard_windows reports for shard k.  `windows` are in chromosome coordinates.  Every context launches its
 * pass before any result is waited for; out_host receives n_windows records in the order of `windows`, byte for
 * byte what one context holding the whole matrix returns. */
int impop_scan_sharded(impop_ctx *const *ctxs, const impop_matrix *const *slabs, const uint64_t *slab_site_begin, int n_ctx,
                       const impop_window *windows, uint64_t n_windows, const uint64_t *mask_p, const uint64_t *mask_a,
                       const uint64_t *mask_b, const impop_scan_params *params, impop_window_stats *out_host);

/* One process per GPU: a communicator over RCCL (xGMI inside a node).  Rank 0 calls impop_comm_unique_id and
 * hands the 128 bytes to the other ranks by any out-of-band means (file, MPI, the launcher's store); then every
 * rank calls impop_comm_create with its context.  RCCL is loaded (dlopen librccl.so.1) by these two calls only. */
#define IMPOP_COMM_ID_BYTES 128
typedef struct impop_comm impop_comm;
int impop_comm_unique_id(void *id_out /* IMPOP_COMM_ID_BYTES */);
int impop_comm_create(impop_ctx *ctx, const void *unique_id, int world, int rank, impop_comm **out);
int impop_comm_destroy(impop_comm *comm);
/* ncclAllGather of bytes_per_rank bytes from every rank into d_all (world x bytes_per_rank, rank order); device
 * pointers; enqueued on the context's stream behind the scans that wrote d_local — no host synchronisation. */
int impop_gather(impop_comm *comm, const void *d_local, size_t bytes_per_rank, void *d_all);
/* The scan's exchange step in one call: this rank's records (device pointer, the shard impop_shard_range gives
 * this rank out of n_total_windows; e.g. impop_scan_plan_device_records) are all-gathered and every rank
 * receives all n_total_windows records in global window order on the host.  Synchronises the stream. */
int impop_gather_records(impop_comm *comm, const void *d_local_records, uint64_t n_total_windows,
                         impop_window_stats *out_host);
/* The one reduction of the path: a single giant window's Gram matrix, site axis split over
 * ranks — in-place ncclAllReduce(sum) of `count` int64 on the device, on the context's stream. */
int impop_allreduce_i64(impop_comm *comm, int64_t *d_values, size_t count);

/* ---- all-pairs path -------------------------------------------------------
 * I_ij = #sites of the window carried by both i and j (the quantity behind
 * `impg similarity`'s estimated.identity, run_pica2_impg.sh:162); a_i = I_ii.
 * out: int32 [n_hap * n_hap] row-major on the host. */
int impop_pairwise_counts(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                          int32_t *out_host);

#define IMPOP_IDENTITY_MATCH 0 /* (W - H_ij)/W, H = a_i + a_j - 2 I_ij */
#define IMPOP_IDENTITY_DICE 1  /* 2 I_ij / (a_i + a_j); 0/0 -> 1.0 */

/* Identity matrix of a window as doubles (what a .sim file would hold). */
int impop_pairwise_identity(impop_ctx *ctx, const impop_matrix *m, uint64_t site_begin, uint64_t site_end,
                            int identity_kind, double *out_host);

/* Full pica2 / h-fst / af semantics (thresholds, rounding, grouping) for a
 * batch of windows straight from the bit matrix; identity never leaves the GPU. */
typedef struct impop_pairwise_params {
    uint32_t struct_size;
    int32_t identity_kind;   /* IMPOP_IDENTITY_* */
    double threshold;        /* pica2 -t (pica2.py:175) */
    int32_t round_digits;    /* pica2 -r / h-fst -r; < 0 = none */
    int32_t d_pi_mode;       /* as impop_scan_params */
    int32_t s_scope;         /* as impop_scan_params; 2 = S and Tajima's D not needed: skips the site scan (s_all = s_p = 0, tajima_d = NaN) */
    uint32_t fst_method;     /* 0 = h-fst.py / hud.py direct; 1 = hud.py -m grouped at `threshold` (hud.py:64-128, 235-263) */
} impop_pairwise_params;
typedef struct impop_pairwise_stats { /* 96 bytes */
    double pi, pi_site;                      /* pica2.py:154,164 on subset P with grouping */
    double fst, pi_a, pi_b, pi_xy, dxy, da;  /* h-fst.py:233-249 */
    double tajima_d;                         /* tj_d.py:47 wired per d_pi_mode / s_scope */
    uint32_t n_groups;                       /* pica2.py:114 */
    uint32_t s_all, s_p, n_sites;
    uint64_t reserved;
} impop_pairwise_stats;
int impop_pairwise_scan(impop_ctx *ctx, const impop_matrix *m, const impop_window *windows, uint64_t n_windows,
                        const uint64_t *mask_p, const uint64_t *mask_a, const uint64_t *mask_b,
                        const impop_pairwise_params *params, impop_pairwise_stats *out_host);
/* Measurement aid (like impop_scan_plan_timing): with timing enabled every Gram launch of impop_pairwise_scan on this context
 * is bracketed with hipEvents on the context's stream; elapsed() synchronises and returns the summed Gram-kernel time and the
 * number of launches since enable / reset. */
/* Test aid: ORs `bits` into the context's device error word, as a kernel whose consistency check trips would; the next call
 * that checks the word (impop_pairwise_scan, impop_pi_from_identity, impop_fst_grouped_from_identity) returns
 * IMPOP_E_INTERNAL and clears it. */
int impop_debug_raise_device_error(impop_ctx *ctx, uint32_t bits);
int impop_ctx_gram_timing(impop_ctx *ctx, int enable);
int impop_ctx_gram_elapsed(impop_ctx *ctx, double *total_ms, uint64_t *launches);

/* impop_pairwise_scan over several devices, sharded like impop_scan_sharded (declared with the multi-GPU entry points
 * above; run_pica2_impg.sh:125-236 / run_h-fst.sh:155-190 with thresholds):
 * shard k = the windows impop_shard_windows gives it, contracted on ctxs[k] from slabs[k] (which needs its hap-major
 * operand, IMPOP_KEEP_HAP_MAJOR).  Every shard runs on a host thread of its own, so the devices work side by side;
 * contexts must be distinct.  out_host: n_windows records in the order of `windows`, byte for byte what one context
 * holding the whole matrix returns. */
int impop_pairwise_scan_sharded(impop_ctx *const *ctxs, const impop_matrix *const *slabs, const uint64_t *slab_site_begin,
                                int n_ctx, const impop_window *windows, uint64_t n_windows, const uint64_t *mask_p,
                                const uint64_t *mask_a, const uint64_t *mask_b, const impop_pairwise_params *params,
                                impop_pairwise_stats *out_host);


/* ---- statistics on a given identity matrix (the .sim drop-in path) ---------
 * These take what read_similarity_file (pica2.py:6-58, h-fst.py:84-119) yields,
 * densified by the caller, and run the reference's arithmetic on the GPU. */

/* pica2.analyze_similarity_matrix (pica2.py:60-169).  seq_len 0 = None.
 * group_of (nullable, n entries): 0-based index of each element's group in the
 * reference's sorted group order (pica2.py:110-112).
 * seed_rank (nullable, n entries, distinct values): the order in which the greedy grouping of
 * pica2.py:96-110 takes its seeds.  The reference takes them with set.pop() from `remaining =
 * set(elements)`, which walks the set's hash table from slot 0 without ever rehashing, so its seed order is
 * the iteration order of `set(elements)` restricted to what is left; a caller in the same interpreter passes
 * seed_rank[i] = position of element i in list(set(elements)) and gets the reference's groups for
 * non-transitive tables too (impop_amd/pica2.py does).  NULL = seed with the smallest remaining index
 * (lexicographically smallest name): deterministic, and one of the orders the reference can take.
 * detail (nullable): the two intermediate values pica2's log prints (pica2.py:158-159). */
typedef struct impop_pica2_detail {
    double sum_2pairs;          /* sum(2 * pair for pair in group_pairs) */
    uint64_t n_pairs_with_data; /* len(group_pairs) */
} impop_pica2_detail;
int impop_pi_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, double threshold, int round_digits,
                           uint64_t seq_len, const uint32_t *seed_rank, double *pi, double *pi_site, uint32_t *group_of,
                           uint32_t *n_groups, impop_pica2_detail *detail);
/* The "Step 2" table of pica2's log (pica2.py:125-145) for given groups: rep[g] = index of group g's first
 * member, group_size[g]; for the pairs g < h in row-major order sims_out = identity of the two
 * representatives (rounded like the analysis; NaN = pair absent) and values_out = (1 - sim) * f_g * f_h.
 * Both arrays hold n_groups*(n_groups-1)/2 doubles. */
int impop_pica2_pair_terms(impop_ctx *ctx, const double *ident, uint32_t n, int round_digits, const uint32_t *rep,
                           const uint32_t *group_size, uint32_t n_groups, double *sims_out, double *values_out);

/* h-fst.calculate_fst (h-fst.py:173-249).  in_a / in_b: n membership flags.
 * out[6] = fst, pi_a, pi_b, pi_xy, dxy, da.
 * counts[6] = pairs_a, missing_a, pairs_b, missing_b, pairs_between, missing_between. */
int impop_fst_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, const uint8_t *in_a,
                            const uint8_t *in_b, uint64_t seq_len, int round_digits, double *out,
                            uint64_t *counts);

/* scripts/hudson/hud.py calculate_fst(method='grouped') (hud.py:64-128, 173-300): greedy groups
 * inside each population at `threshold`, frequency-weighted group-pair sums; the similarity of two
 * groups is the first pair (members in sorted order) present in the table.  out[6] as above;
 * counts[6] = groups_a, missing_a, groups_b, missing_b, group pairs between, missing between.
 * seed_rank (nullable, n entries): seed order of hud.py:64-86's set.pop() inside each population, as for
 * impop_pi_from_identity — values must be distinct among the members of A and among those of B (position
 * in list(set(pop_a)) / list(set(pop_b))); entries of non-members are ignored. */
int impop_fst_grouped_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, const uint8_t *in_a,
                                    const uint8_t *in_b, double threshold, uint64_t seq_len, int round_digits,
                                    const uint32_t *seed_rank, double *out, uint64_t *counts);

/* tj_d.tajimas_d (tj_d.py:47-69) for `count` (n, S, pi) triples.  comps
 * (nullable): count x 10 doubles a1,a2,b1,b2,c1,c2,e1,e2,numerator,denominator.
 * Returns IMPOP_E_INVALID (message = the reference's ValueError text) if any
 * triple has n < 2, S < 0 or pi < 0; nothing is written in that case. */
int impop_tajimas_d(impop_ctx *ctx, const int64_t *n, const double *S, const double *pi, uint64_t count,
                    double *D, double *comps);

/* af.cluster (af.py:35-44): connected components of {identity >= threshold}
 * ordered by (-size, members); cluster_of[i] = 0-based cluster rank (c1 = 0),
 * sizes (nullable): n entries, first n_clusters valid. */
int impop_cluster_from_identity(impop_ctx *ctx, const double *ident, uint32_t n, double threshold,
                                uint32_t *cluster_of, uint32_t *n_clusters, uint32_t *sizes);

/* ---- native .sim ingest (host code) ------------------------------------------
 * Replaces pica2.read_similarity_file (pica2.py:6-58) / h-fst.read_similarity_file
 * (h-fst.py:84-119) for clean tab-separated files.  flavor 0 = pica2 (the first
 * unparsable value stops the parse: bad_line / impop_sim_bad_text report it, the
 * caller prints the reference's message and exits 1), flavor 1 = h-fst (unparsable
 * values are skipped and counted in n_bad).  Returns IMPOP_E_UNSUPPORTED for any file
 * shape it is not certain CPython's csv + float() would read identically (quotes,
 * short rows, unusual number syntax, missing columns): the caller then uses the
 * Python reader.  A missing file returns IMPOP_E_INVALID. */
typedef struct impop_sim impop_sim;
int impop_sim_parse(const char *path, int flavor, impop_sim **out);
int impop_sim_info(const impop_sim *s, uint32_t *n_names, uint64_t *n_rows, uint64_t *names_bytes,
                   int64_t *bad_line, uint64_t *n_bad);
int impop_sim_names(const impop_sim *s, char *buf);          /* NUL-separated, sorted */
/* first_seen_out[k] (n_names entries) = sorted rank of the k-th distinct name in file order (group.a before
 * group.b within a row): the insertion order of the reference reader's `elements` set (pica2.py:45-46),
 * from which the caller rebuilds that set and hence pica2's seed order (see impop_pi_from_identity). */
int impop_sim_first_seen(const impop_sim *s, uint32_t *first_seen_out);
int impop_sim_bad_text(const impop_sim *s, char *buf, size_t buflen);
int impop_sim_dense(const impop_sim *s, double *out);        /* n x n, sorted-name order, NaN = absent */
int impop_sim_free(impop_sim *s);

/* ---- native GFA ingest (host code) ---------------------------------------------
 * S / P / W lines of the window graph (`impg query -o gfa`, run_tajd.sh:126; `odgi view -g`, :140) -> the NODE-level
 * presence matrix: rows = paths and walks sorted by name (W named sample#hap#seqid[:start-end]), columns = segments
 * (decimal ids in numeric order, then the others), bit-packed hap-major as impop_matrix_upload takes it; the segment
 * lengths are the site weights of impop_matrix_set_site_weights.  ref_prefix (nullable): the first path whose name
 * starts with it gives every column a reference coordinate (start parsed from a trailing ":start-end"; columns off
 * the reference inherit the preceding one; non-decreasing).  Same rules as impop_amd/extract.py:from_gfa
 * (expand_bp=False), which the caller falls back to on ANY non-zero status (so error texts stay Python's). */
typedef struct impop_gfa impop_gfa;
int impop_gfa_parse(const char *path, const char *ref_prefix, impop_gfa **out);
int impop_gfa_info(const impop_gfa *g, uint32_t *n_path, uint64_t *n_seg, uint64_t *names_bytes, int64_t *ref_row);
int impop_gfa_names(const impop_gfa *g, char *buf);                       /* NUL-separated, row order */
int impop_gfa_bits(const impop_gfa *g, uint64_t *bits_hap_major, uint64_t row_stride_words);
int impop_gfa_lengths(const impop_gfa *g, uint32_t *lengths);            /* per column */
int impop_gfa_positions(const impop_gfa *g, int64_t *positions);         /* per column; needs ref_prefix */
int impop_gfa_free(impop_gfa *g);
/* `odgi paths -H` table (header row; path.name, path.length, node.count; then one 0 / visit-count column per node — the
 * shape scripts/wip/op-afs.py:112 reads) -> the same handle: rows sorted by name, one column per node (all lengths 1,
 * no positions).  Rows are parsed by several host threads.  Same rules as impop_amd/extract.py:from_paths_table, which
 * the caller falls back to on ANY non-zero status. */
int impop_paths_table_parse(const char *path, impop_gfa **out);

/* CPython round(x, ndigits) (pica2.py:83, h-fst.py:150) evaluated on the GPU,
 * exposed so that the device implementation can be fuzzed against CPython. */
int impop_py_round(impop_ctx *ctx, const double *x, uint64_t count, int ndigits, double *out);

#ifdef __cplusplus
}
#endif
#endif /* IMPOP_HIP_H */
