/* Plain-C caller of the multi-GPU entries of include/impop_hip.h: no Python, no torch.
 *
 *   gcc -O2 -std=c99 -Iinclude examples/scan_sharded_from_c.c -o scan_sharded_from_c -Limpop_amd -limpop_hip \
 *       -Wl,-rpath,$PWD/impop_amd
 *   ./scan_sharded_from_c [n_shards]
 *
 * What the serial window loops of run_tajd.sh:103-196 / run_h-fst.sh:155-190 become on a multi-GPU node: the
 * window list is cut into contiguous ranges (impop_shard_windows), every shard's slab of sites is uploaded to its
 * own context — one per device when the machine has as many devices as shards, otherwise several contexts share
 * device 0 — impop_scan_sharded runs all of them and returns the records in window order; impop_pairwise_scan_sharded
 * does the same for the all-pairs mode (thresholded pica2 + h-fst).  Both results are compared byte for byte with one
 * context holding the whole matrix, and the scan records are pushed through the one-process-per-GPU exchange
 * (impop_comm_* + impop_gather_records over RCCL) with a one-rank communicator. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "impop_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != IMPOP_OK) {                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, impop_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

enum { N_HAP = 40, N_SITE = 6000, WORDS = (N_SITE + 63) / 64, WIN = 500, STEP = 250, MAX_SHARDS = 8 };

static uint64_t bits[N_HAP][WORDS];

int main(int argc, char **argv) {
    int n_shards = argc > 1 ? atoi(argv[1]) : 3;
    if (n_shards < 1 || n_shards > MAX_SHARDS) return 2;
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (int s = 0; s < N_SITE; ++s) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int anc = (int)(x & 1), split = (x >> 8) % 40 == 0;
        for (int h = 0; h < N_HAP; ++h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            if (anc ^ (split && (h % 3 == 0)) ^ ((x >> 20) % 150 == 0)) bits[h][s >> 6] |= 1ull << (s & 63);
        }
    }
    /* sliding windows (50 % overlap): neighbouring slabs share a halo */
    impop_window win[64];
    uint64_t n_win = 0;
    for (uint64_t b = 0; b + WIN <= N_SITE; b += STEP) {
        win[n_win].site_begin = b; win[n_win].site_end = b + WIN; win[n_win].seq_len = WIN; ++n_win;
    }
    const uint64_t mask_a = 0x00000FFFFFull, mask_b = 0xFFFFF00000ull;

    /* reference: one context, whole matrix */
    impop_ctx *ctx0 = NULL;
    impop_matrix *whole = NULL;
    static impop_window_stats want[64], got[64], gathered[64];
    CHECK(impop_ctx_create(0, NULL, &ctx0));
    CHECK(impop_matrix_upload(ctx0, &bits[0][0], N_HAP, N_SITE, WORDS, IMPOP_KEEP_SITE_BLOCKED | IMPOP_KEEP_HAP_MAJOR, &whole));
    CHECK(impop_scan(ctx0, whole, win, n_win, NULL, &mask_a, &mask_b, NULL, want));

    /* shards: a context + the slab its windows touch, per shard */
    int n_dev = 0;
    CHECK(impop_device_count(&n_dev));
    impop_ctx *ctxs[MAX_SHARDS];
    impop_matrix *slabs[MAX_SHARDS];
    uint64_t slab_begin[MAX_SHARDS];
    for (int k = 0; k < n_shards; ++k) {
        uint64_t first, cnt, s0, s1;
        CHECK(impop_shard_windows(win, n_win, n_shards, k, &first, &cnt, &s0, &s1));
        CHECK(impop_ctx_create(n_dev >= n_shards ? k : 0, NULL, &ctxs[k]));
        /* slab = whole 64-bit words of the hap-major rows around [s0, s1) */
        const uint64_t w0 = s0 / 64, w1 = (s1 + 63) / 64 > w0 ? (s1 + 63) / 64 : w0 + 1;
        uint64_t n_slab_site = (w1 * 64 < N_SITE ? w1 * 64 : N_SITE) - w0 * 64;
        CHECK(impop_matrix_upload(ctxs[k], &bits[0][w0], N_HAP, n_slab_site, WORDS, IMPOP_KEEP_SITE_BLOCKED | IMPOP_KEEP_HAP_MAJOR,
                                  &slabs[k]));
        slab_begin[k] = w0 * 64;
        printf("shard %d on device %d: windows [%llu, %llu), sites [%llu, %llu)\n", k, n_dev >= n_shards ? k : 0,
               (unsigned long long)first, (unsigned long long)(first + cnt), (unsigned long long)s0, (unsigned long long)s1);
    }
    CHECK(impop_scan_sharded(ctxs, (const impop_matrix *const *)slabs, slab_begin, n_shards, win, n_win, NULL, &mask_a, &mask_b,
                             NULL, got));
    const int same = memcmp(want, got, n_win * sizeof want[0]) == 0;
    printf("%llu windows over %d shards: records %s one context's\n", (unsigned long long)n_win, n_shards,
           same ? "byte-identical to" : "DIFFER from");

    /* the all-pairs mode (pica2 -t 0.999 -r 5 + h-fst per window) over the same shards */
    static impop_pairwise_stats pw_want[64], pw_got[64];
    impop_pairwise_params pp;
    memset(&pp, 0, sizeof pp);
    pp.struct_size = sizeof pp; pp.identity_kind = IMPOP_IDENTITY_MATCH; pp.threshold = 0.999; pp.round_digits = 5;
    CHECK(impop_pairwise_scan(ctx0, whole, win, n_win, NULL, &mask_a, &mask_b, &pp, pw_want));
    CHECK(impop_pairwise_scan_sharded(ctxs, (const impop_matrix *const *)slabs, slab_begin, n_shards, win, n_win, NULL, &mask_a,
                                      &mask_b, &pp, pw_got));
    const int same3 = memcmp(pw_want, pw_got, n_win * sizeof pw_want[0]) == 0;
    printf("all-pairs mode over %d shards: records %s one context's (window 0: %u groups, pi %.8f)\n", n_shards,
           same3 ? "byte-identical to" : "DIFFER from", pw_got[0].n_groups, pw_got[0].pi_site);

    /* the one-process-per-GPU exchange with the one rank this process is */
    unsigned char id[IMPOP_COMM_ID_BYTES];
    impop_comm *comm = NULL;
    impop_scan_plan *plan = NULL;
    void *d_rec = NULL;
    CHECK(impop_comm_unique_id(id));
    CHECK(impop_comm_create(ctx0, id, 1, 0, &comm));
    CHECK(impop_scan_plan_create(ctx0, whole, win, n_win, NULL, &mask_a, &mask_b, NULL, &plan));
    CHECK(impop_scan_plan_launch(plan, NULL));
    CHECK(impop_scan_plan_device_records(plan, &d_rec));
    CHECK(impop_gather_records(comm, d_rec, n_win, gathered));
    const int same2 = memcmp(want, gathered, n_win * sizeof want[0]) == 0;
    printf("RCCL all-gather (1 rank): records %s\n", same2 ? "intact" : "DIFFER");
    CHECK(impop_scan_plan_destroy(plan));
    CHECK(impop_comm_destroy(comm));
    for (int k = 0; k < n_shards; ++k) {
        CHECK(impop_matrix_free(ctxs[k], slabs[k]));
        CHECK(impop_ctx_destroy(ctxs[k]));
    }
    CHECK(impop_matrix_free(ctx0, whole));
    CHECK(impop_ctx_destroy(ctx0));
    return same && same2 && same3 ? 0 : 1;
}
