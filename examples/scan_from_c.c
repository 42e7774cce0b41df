/* Plain-C caller of the drop-in boundary (include/impop_hip.h): no Python, no torch.
 *
 *   gcc -O2 -Iinclude examples/scan_from_c.c -o scan_from_c -Limpop_amd -limpop_hip -Wl,-rpath,$PWD/impop_amd
 *   ./scan_from_c
 *
 * Uploads a small haplotype x site matrix (hap-major bits), scans three windows for pi / Hudson Fst /
 * Tajima's D / S in one pass and computes Tajima's D for the known answer of doc/how_tjd.md:45. */
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

int main(void) {
    enum { N_HAP = 12, N_SITE = 1000, WORDS = (N_SITE + 63) / 64 };
    static uint64_t bits[N_HAP][WORDS];
    uint64_t x = 88172645463325252ull; /* xorshift: haplotypes 0-5 and 6-11 are two noisy lineages */
    for (int s = 0; s < N_SITE; ++s) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int anc = (int)(x & 1), split = (x >> 8) % 50 == 0;
        for (int h = 0; h < N_HAP; ++h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            int b = anc ^ (split && h >= 6) ^ ((x >> 20) % 200 == 0);
            if (b) bits[h][s >> 6] |= 1ull << (s & 63);
        }
    }
    impop_ctx *ctx = NULL;
    impop_matrix *m = NULL;
    CHECK(impop_ctx_create(0, NULL, &ctx));
    CHECK(impop_matrix_upload(ctx, &bits[0][0], N_HAP, N_SITE, WORDS, IMPOP_KEEP_SITE_BLOCKED, &m));
    const impop_window win[3] = {{0, 1000, 1000}, {0, 500, 500}, {250, 750, 0}};
    const uint64_t mask_a = 0x03Full, mask_b = 0xFC0ull; /* haplotypes 0-5 vs 6-11 */
    impop_window_stats rec[3];
    CHECK(impop_scan(ctx, m, win, 3, NULL, &mask_a, &mask_b, NULL, rec));
    for (int i = 0; i < 3; ++i)
        printf("window [%llu,%llu): S=%u pi=%.6g pi_site=%.6g fst=%.6g dxy=%.6g D=%.6g\n", (unsigned long long)win[i].site_begin,
               (unsigned long long)win[i].site_end, rec[i].s_all, rec[i].pi, rec[i].pi_site, rec[i].fst, rec[i].dxy, rec[i].tajima_d);
    int64_t n = 446;
    double S = 20.0, pi = 0.59146123, D = 0.0;
    CHECK(impop_tajimas_d(ctx, &n, &S, &pi, 1, &D, NULL));
    printf("tajimas_d(446, 20, 0.59146123) = %.10f\n", D);
    CHECK(impop_matrix_free(ctx, m));
    CHECK(impop_ctx_destroy(ctx));
    return 0;
}
