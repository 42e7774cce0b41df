/*
 * impop_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the pairwise-diversity hot path of pangenome/impop
 * (scripts/pica2.py, scripts/h-fst.py, scripts/tj_d.py, scripts/af.py, and the
 * extras scripts/hudson/hud.py grouped Fst and scripts/wip/ehhgfa.py EHH).  Every
 * function cites the reference file:line it follows.  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check /
 * time the HIP engine against an independent implementation.  Nothing under
 * impop_amd/ or scripts/ may import, link or execute this file.
 *
 * Parity status: PINNED.  The reference is pure-stdlib Python and importable in
 * the build container; oracle/gen_golden.py runs the real reference functions
 * on seeded inputs and stores their full-precision outputs in tests/golden/;
 * tests/test_oracle_golden.py checks this file against those vectors and the
 * known answers listed in SURVEY.md §4 (pica2, h-fst, tj_d, af, hud grouped and
 * calc_EHH all have captured vectors).  Not pinned (third-party tools absent
 * from the reference tree: impg/odgi/povu): bit-matrix -> identity and the
 * segregating-site count S, which this engine *defines* (see oracle_* "engine
 * definition" comments) — parity with the reference is claimed from the
 * identity matrix onward.
 *
 * Conventions: haplotypes are indices 0..n-1 in *lexicographic name order* (the
 * host sorts names first), so "sorted(group)", "groups.sort()" and "first
 * member" of the reference all reduce to index order.  Dense n×n identity
 * matrices are row-major doubles; NaN marks a pair absent from the .sim table.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile) — no FMA
 * contraction so the fp64 operation order below is exactly CPython's.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* CPython round(x, ndigits) — pica2.py:83, h-fst.py:150,163.                 */
/* CPython (Objects/floatobject.c double_round) formats the exact binary value */
/* with ndigits decimals (correctly rounded, ties-to-even on the exact value)  */
/* and parses the string back.  glibc's printf/strtod are both exact, so this  */
/* is the same function (checked against CPython in tests/test_oracle_golden).*/
ORACLE_API double oracle_py_round(double x, int ndigits) {
    if (!isfinite(x)) return x;
    if (ndigits > 22) ndigits = 22; /* beyond this CPython switches algorithm; never used by impop */
    if (ndigits < 0) ndigits = 0;
    char buf[512];
    snprintf(buf, sizeof buf, "%.*f", ndigits, x);
    return strtod(buf, NULL);
}

/* ------------------------------------------------------------------------- */
/* tj_d.py:41-69                                                              */
/* comps[10] = a1,a2,b1,b2,c1,c2,e1,e2,numerator,denominator (tj_d.py:28-39). */
/* returns 0, or -1 for the ValueError cases (tj_d.py:48-51).                 */
ORACLE_API int oracle_tajimas_d(int64_t n, double S, double pi, double *D, double *comps) {
    if (n < 2) return -1;          /* tj_d.py:48-49 */
    if (S < 0 || pi < 0) return -1; /* tj_d.py:50-51 */
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = 1; i < n; ++i) a1 += 1.0 / (double)i;                 /* tj_d.py:41-42 (int 0 + float…) */
    for (int64_t i = 1; i < n; ++i) a2 += 1.0 / ((double)i * (double)i);   /* tj_d.py:44-45 (i*i exact < 2^53) */
    double dn = (double)n;
    double b1 = (dn + 1.0) / (3.0 * (dn - 1.0));                           /* :55 */
    double b2 = 2.0 * (dn * dn + dn + 3.0) / (9.0 * dn * (dn - 1.0));      /* :56 */
    double c1 = b1 - (1.0 / a1);                                           /* :57 */
    double c2 = b2 - ((dn + 2.0) / (a1 * dn)) + (a2 / (a1 * a1));          /* :58 */
    double e1 = c1 / a1;                                                   /* :59 */
    double e2 = c2 / (a1 * a1 + a2);                                       /* :60 */
    double num = pi - (S / a1);                                            /* :62 */
    double den = (S > 0) ? sqrt(e1 * S + e2 * S * (S - 1.0)) : NAN;        /* :63 */
    /* :65  `denominator and not math.isclose(denominator, 0.0)`: NaN is truthy;
       isclose(x, 0.0) with rel_tol=1e-9, abs_tol=0 is true only for x == 0.   */
    double d;
    if (den != 0.0) d = num / den; /* NaN den -> NaN */
    else d = NAN;
    if (D) *D = d;
    if (comps) {
        comps[0] = a1; comps[1] = a2; comps[2] = b1; comps[3] = b2; comps[4] = c1;
        comps[5] = c2; comps[6] = e1; comps[7] = e2; comps[8] = num; comps[9] = den;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Greedy grouping of pica2.py:94-112 and hud.py:64-86, restated literally:    */
/*   remaining = set(elements); while remaining: current = remaining.pop();    */
/*   every `other` still remaining with sim(current, other) > threshold joins; */
/*   groups.append(sorted(group)); finally groups.sort().                      */
/* set.pop() walks the hash table from slot 0 and removals never rehash, so    */
/* the seeds come in the iteration order of `set(elements)` restricted to what */
/* is left.  seed_rank (nullable): seed_rank[idx[k]] = position of member k in */
/* that order (distinct among the members); NULL = index order (the engine's   */
/* documented default: smallest remaining name).  idx (nullable) = members.    */
/* grp[k] = group of member k, groups numbered as after groups.sort() (by      */
/* their smallest member); rep[g] (nullable) = member position of groups[g][0] */
static uint32_t greedy_groups(const double *sim, uint32_t n, const uint32_t *idx, uint32_t m, double thr, int rd,
                              const uint32_t *seed_rank, uint32_t *grp, uint32_t *rep) {
    const uint32_t NONE = 0xFFFFFFFFu;
    uint32_t *order = (uint32_t *)malloc((size_t)(m ? m : 1) * sizeof(uint32_t));
    for (uint32_t k = 0; k < m; ++k) order[k] = k;
    if (seed_rank) /* insertion sort by rank: test sizes are small */
        for (uint32_t k = 1; k < m; ++k) {
            uint32_t v = order[k], j = k;
            while (j > 0 && seed_rank[idx ? idx[order[j - 1]] : order[j - 1]] > seed_rank[idx ? idx[v] : v]) { order[j] = order[j - 1]; --j; }
            order[j] = v;
        }
    uint32_t *raw = (uint32_t *)malloc((size_t)(m ? m : 1) * sizeof(uint32_t));
    uint32_t *gmin = (uint32_t *)malloc((size_t)(m ? m : 1) * sizeof(uint32_t));
    for (uint32_t k = 0; k < m; ++k) raw[k] = NONE;
    uint32_t G = 0;
    for (uint32_t q = 0; q < m; ++q) {
        const uint32_t s = order[q];            /* remaining.pop() */
        if (raw[s] != NONE) continue;
        raw[s] = G; gmin[G] = s;
        for (uint32_t o = 0; o < m; ++o) {      /* for other in list(remaining) */
            if (raw[o] != NONE) continue;
            uint32_t a = idx ? idx[s] : s, b = idx ? idx[o] : o;
            double v = a <= b ? sim[(size_t)a * n + b] : sim[(size_t)b * n + a];
            if (isnan(v)) continue;             /* pair absent: get() is None / key not in similarities */
            if (rd >= 0) v = oracle_py_round(v, rd);
            if (v > thr) { raw[o] = G; if (o < gmin[G]) gmin[G] = o; } /* strict > (pica2.py:106, hud.py:80) */
        }
        ++G;
    }
    /* sorted(group) puts the smallest member first; groups.sort() orders the groups by it */
    uint32_t *newid = (uint32_t *)malloc((size_t)(G ? G : 1) * sizeof(uint32_t));
    for (uint32_t g = 0; g < G; ++g) {
        uint32_t r = 0;
        for (uint32_t h = 0; h < G; ++h) r += gmin[h] < gmin[g];
        newid[g] = r;
        if (rep) rep[r] = gmin[g];
    }
    for (uint32_t k = 0; k < m; ++k) grp[k] = newid[raw[k]];
    free(order); free(raw); free(gmin); free(newid);
    return G;
}

/* ------------------------------------------------------------------------- */
/* pica2.py:60-169  analyze_similarity_matrix                                  */
/* sim: dense n×n (symmetric; only (min,max) entry is consulted like the dict  */
/* key at pica2.py:86); NaN = missing.  round_digits < 0 => None.              */
/* Seed rule: the reference pops an arbitrary set element (pica2.py:100); the  */
/* engine fixes seed = smallest remaining index (= lexicographically smallest  */
/* name).  For inputs where "> threshold" is an equivalence relation every      */
/* seed order gives the same groups, so goldens use such inputs.               */
/* group_of (nullable) receives the 0-based group index of every element,      */
/* groups numbered in sorted order (pica2.py:110-112).                          */
ORACLE_API int oracle_pica2_seeded(const double *sim, uint32_t n, double threshold, int round_digits,
                                   double seq_len, const uint32_t *seed_rank, double *pi_out, double *pi_site_out,
                                   uint32_t *group_of, uint32_t *n_groups_out) {
    double *S = NULL;
    if (round_digits >= 0) { /* pica2.py:81-83 */
        S = (double *)malloc((size_t)n * n * sizeof(double));
        for (size_t k = 0; k < (size_t)n * n; ++k) S[k] = isnan(sim[k]) ? sim[k] : oracle_py_round(sim[k], round_digits);
        sim = S;
    }
#define SIM(i, j) ((i) <= (j) ? sim[(size_t)(i) * n + (j)] : sim[(size_t)(j) * n + (i)])
    uint32_t *grp = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t *rep = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t *gsz = (uint32_t *)calloc((size_t)(n ? n : 1), sizeof(uint32_t));
    /* pica2.py:94-112; values are already rounded in place (:81-83), hence rd = -1 here */
    uint32_t G = greedy_groups(sim, n, NULL, n, threshold, -1, seed_rank, grp, rep);
    for (uint32_t i = 0; i < n; ++i) gsz[grp[i]]++;
    double pi = 0.0, pi_site = 0.0;
    int have_pairs = 0;
    uint32_t total = n; /* pica2.py:121 */
    if (total != 0) {
        double acc = 0.0; /* Python sum() starts at int 0; 0 + x == x exactly */
        for (uint32_t i = 0; i < G; ++i)
            for (uint32_t j = i + 1; j < G; ++j) { /* :125-126 */
                double s = SIM(rep[i], rep[j]);    /* :128-131 */
                if (isnan(s)) continue;            /* :132-134 */
                double fi = (double)gsz[i] / (double)total; /* :137 */
                double fj = (double)gsz[j] / (double)total; /* :138 */
                double pv = (1 - s) * fi * fj;              /* :139 */
                acc += 2 * pv;                              /* :154 sum(2*pair ...) */
                have_pairs = 1;
            }
        if (have_pairs) {
            pi = ((double)n / (double)(n - 1)) * acc; /* :154 */
            if (seq_len != 0.0) pi_site = pi / seq_len; /* :163-164 */
            else pi_site = NAN;                          /* None */
        } /* else (0.0, 0.0) :150-152 */
    }     /* else (0.0, 0.0) :122-124 */
#undef SIM
    if (pi_out) *pi_out = pi;
    if (pi_site_out) *pi_site_out = pi_site;
    if (group_of) memcpy(group_of, grp, (size_t)n * sizeof(uint32_t));
    if (n_groups_out) *n_groups_out = G;
    free(grp); free(rep); free(gsz); free(S);
    return 0;
}
/* seed = smallest remaining index (lexicographically smallest name) */
ORACLE_API int oracle_pica2(const double *sim, uint32_t n, double threshold, int round_digits,
                            double seq_len, double *pi_out, double *pi_site_out,
                            uint32_t *group_of, uint32_t *n_groups_out) {
    return oracle_pica2_seeded(sim, n, threshold, round_digits, seq_len, NULL, pi_out, pi_site_out, group_of, n_groups_out);
}

/* ------------------------------------------------------------------------- */
/* h-fst.py:130-171 calculate_diversity.  in1/in2: membership flags (0/1).     */
/* in2 == NULL => within in1.  Iteration order = index order (the reference    */
/* iterates Python sets, order arbitrary; affects only the last bits).         */
static void hfst_diversity(const double *sim, uint32_t n, const uint8_t *in1, const uint8_t *in2,
                           int round_digits, double *mean, uint64_t *count, uint64_t *missing) {
    double acc = 0.0; uint64_t cnt = 0, miss = 0;
#define SIM(i, j) ((i) <= (j) ? sim[(size_t)(i) * n + (j)] : sim[(size_t)(j) * n + (i)])
    if (!in2) {
        for (uint32_t i = 0; i < n; ++i) { if (!in1[i]) continue;
            for (uint32_t j = i + 1; j < n; ++j) { if (!in1[j]) continue; /* :142-143 */
                double s = SIM(i, j);
                if (isnan(s)) { ++miss; continue; }                  /* :152-153 */
                if (round_digits >= 0) s = oracle_py_round(s, round_digits); /* :149-150 */
                acc += (1 - s); ++cnt;                                /* :151 */
            } }
    } else {
        for (uint32_t i = 0; i < n; ++i) { if (!in1[i]) continue;
            for (uint32_t j = 0; j < n; ++j) { if (!in2[j]) continue; /* :156-157 */
                double s = SIM(i, j);
                if (isnan(s)) { ++miss; continue; }
                if (round_digits >= 0) s = oracle_py_round(s, round_digits);
                acc += (1 - s); ++cnt;
            } }
    }
#undef SIM
    *mean = cnt ? acc / (double)cnt : 0.0; /* :168-171 */
    *count = cnt; *missing = miss;
}

/* h-fst.py:173-249 calculate_fst.  out[6] = fst,pi_a,pi_b,pi_xy,dxy,da;       */
/* counts[6] = count_a,miss_a,count_b,miss_b,count_between,miss_between.      */
ORACLE_API int oracle_hfst(const double *sim, uint32_t n, const uint8_t *in_a, const uint8_t *in_b,
                           double seq_len, int round_digits, double *out, uint64_t *counts) {
    uint8_t *a = (uint8_t *)malloc(n ? n : 1), *b = (uint8_t *)malloc(n ? n : 1);
    for (uint32_t i = 0; i < n; ++i) { /* :181-185 overlap removed from both */
        int ov = in_a[i] && in_b[i];
        a[i] = in_a[i] && !ov; b[i] = in_b[i] && !ov;
    }
    double pi_a, pi_b, dxy; uint64_t c[6];
    hfst_diversity(sim, n, a, NULL, round_digits, &pi_a, &c[0], &c[1]); /* :197 */
    hfst_diversity(sim, n, b, NULL, round_digits, &pi_b, &c[2], &c[3]); /* :200 */
    double pi_xy = 0.5 * (pi_a + pi_b);                                  /* :203 */
    hfst_diversity(sim, n, a, b, round_digits, &dxy, &c[4], &c[5]);      /* :209 */
    double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;                  /* :214-221 */
    if (seq_len > 0) { /* :225-240 (fst not divided) */
        out[0] = fst; out[1] = pi_a / seq_len; out[2] = pi_b / seq_len; out[3] = pi_xy / seq_len;
        out[4] = dxy / seq_len; out[5] = (dxy - pi_xy) / seq_len;
    } else {           /* :241-249 */
        out[0] = fst; out[1] = pi_a; out[2] = pi_b; out[3] = pi_xy; out[4] = dxy; out[5] = dxy - pi_xy;
    }
    if (counts) memcpy(counts, c, sizeof c);
    free(a); free(b);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* af.py:21-44 cluster: connected components of {(a,b): sim >= threshold},     */
/* ordered by (-size, sorted members) (af.py:43) == (-size, min index) because */
/* components are disjoint.  cluster_of[i] = 0-based rank (c1 -> 0).           */
/* NaN (missing row) never links.  Names are truncated at ':' by the caller    */
/* (af.py:13-14).                                                              */
static uint32_t uf_find(uint32_t *p, uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
ORACLE_API int oracle_af_cluster(const double *sim, uint32_t n, double threshold,
                                 uint32_t *cluster_of, uint32_t *n_clusters, uint32_t *sizes) {
    uint32_t *p = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) p[i] = i;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = i; j < n; ++j) {
            double v = sim[(size_t)i * n + j];
            if (!isnan(v) && v >= threshold) { /* af.py:38 non-strict */
                uint32_t ra = uf_find(p, i), rb = uf_find(p, j);
                if (ra != rb) p[rb] = ra;
            }
        }
    uint32_t *size = (uint32_t *)calloc((size_t)(n ? n : 1), sizeof(uint32_t));
    uint32_t *minm = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) minm[i] = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) { uint32_t r = uf_find(p, i); size[r]++; if (i < minm[r]) minm[r] = i; }
    uint32_t K = 0;
    for (uint32_t r = 0; r < n; ++r) if (size[r]) ++K;
    /* rank of root r = #roots that sort before it */
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t r = uf_find(p, i), rank = 0;
        for (uint32_t q = 0; q < n; ++q) {
            if (!size[q] || q == r) continue;
            if (size[q] > size[r] || (size[q] == size[r] && minm[q] < minm[r])) ++rank;
        }
        cluster_of[i] = rank;
        if (sizes) sizes[rank] = size[r];
    }
    if (n_clusters) *n_clusters = K;
    free(p); free(size); free(minm);
    return 0;
}

/* ========================================================================= */
/* Bit-matrix side (engine definitions).           */
/* Interchange layout ("hap-major"): bits[i*stride + (s>>6)] bit (s&63) = 1    */
/* iff haplotype i carries the allele at site s.                               */
static inline int bit_at(const uint64_t *bits, uint64_t stride, uint32_t i, uint64_t s) {
    return (int)((bits[(size_t)i * stride + (s >> 6)] >> (s & 63)) & 1u);
}

/* I_ij = #sites in [s0,s1) where both i and j are 1 (A.2); out n×n int64.     */
ORACLE_API int oracle_pairwise_counts(const uint64_t *bits, uint64_t stride, uint32_t n,
                                      uint64_t s0, uint64_t s1, int64_t *I) {
    if (s1 < s0) return -1;
    uint64_t w0 = s0 >> 6, w1 = (s1 + 63) >> 6;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = i; j < n; ++j) {
            int64_t c = 0;
            for (uint64_t w = w0; w < w1; ++w) {
                uint64_t m = ~0ull;
                if (w == (s0 >> 6)) m &= ~0ull << (s0 & 63);
                if (w == ((s1 - 1) >> 6) && (s1 & 63)) m &= ~0ull >> (64 - (s1 & 63));
                if (s1 == s0) m = 0;
                c += __builtin_popcountll(bits[(size_t)i * stride + w] & bits[(size_t)j * stride + w] & m);
            }
            I[(size_t)i * n + j] = c; I[(size_t)j * n + i] = c;
        }
    return 0;
}

/* identity from counts (A.3).  kind 0 = match: (W - H)/W, H = a_i + a_j - 2I; */
/* kind 1 = dice: 2I/(a_i + a_j), 0/0 -> 1.0.  One IEEE division each.         */
ORACLE_API int oracle_identity(const int64_t *I, uint32_t n, uint64_t W, int kind, double *sim) {
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            int64_t ai = I[(size_t)i * n + i], aj = I[(size_t)j * n + j], x = I[(size_t)i * n + j];
            double v;
            if (kind == 0) { int64_t H = ai + aj - 2 * x; v = W ? (double)((int64_t)W - H) / (double)W : 1.0; }
            else { int64_t d = ai + aj; v = d ? (double)(2 * x) / (double)d : 1.0; }
            sim[(size_t)i * n + j] = v;
        }
    return 0;
}

/* Per-site masked counts over [s0,s1) -> integer window record (A.1).         */
/* masks: n-bit little-endian uint64 words (bit i = haplotype i in the set).   */
/* ints[8] = W, S_all, S_P, S_A, S_B, (unused), ...; sums[4] = sumP,sumA,sumB,sumAB */
static inline int mask_has(const uint64_t *m, uint32_t i) { return (int)((m[i >> 6] >> (i & 63)) & 1u); }
ORACLE_API int oracle_site_scan(const uint64_t *bits, uint64_t stride, uint32_t n, uint64_t s0, uint64_t s1,
                                const uint64_t *mp, const uint64_t *ma, const uint64_t *mb,
                                uint32_t *ints, uint64_t *sums) {
    uint32_t nP = 0, nA = 0, nB = 0;
    for (uint32_t i = 0; i < n; ++i) { nP += mask_has(mp, i); nA += mask_has(ma, i); nB += mask_has(mb, i); }
    uint32_t S_all = 0, S_P = 0, S_A = 0, S_B = 0;
    uint64_t sumP = 0, sumA = 0, sumB = 0, sumAB = 0;
    for (uint64_t s = s0; s < s1; ++s) {
        uint32_t c = 0, cP = 0, cA = 0, cB = 0;
        for (uint32_t i = 0; i < n; ++i) {
            int b = bit_at(bits, stride, i, s);
            c += b; cP += b & mask_has(mp, i); cA += b & mask_has(ma, i); cB += b & mask_has(mb, i);
        }
        S_all += (c > 0 && c < n);
        S_P += (cP > 0 && cP < nP);
        S_A += (cA > 0 && cA < nA);
        S_B += (cB > 0 && cB < nB);
        sumP += (uint64_t)cP * (nP - cP);
        sumA += (uint64_t)cA * (nA - cA);
        sumB += (uint64_t)cB * (nB - cB);
        sumAB += (uint64_t)cA * (nB - cB) + (uint64_t)cB * (nA - cA);
    }
    ints[0] = (uint32_t)(s1 - s0); ints[1] = S_all; ints[2] = S_P; ints[3] = S_A; ints[4] = S_B;
    sums[0] = sumP; sums[1] = sumA; sums[2] = sumB; sums[3] = sumAB;
    return 0;
}

/* Window statistics the reference way: bits -> all-pairs Hamming -> `match`   */
/* identity doubles -> pica2 (threshold >= 1: every haplotype its own group,   */
/* pica2.py:94-154) on subset P, h-fst (h-fst.py:173-249) on A,B, Tajima's D   */
/* wired as run_tajd.sh:166-180 (per-site pi through "%.8f" text, S raw).      */
/* dbl[9] = pi, pi_site, pi_a, pi_b, pi_xy, dxy, da, fst, tajima_d.            */
/* d_pi_mode: 0 reference wiring (8-decimal per-site pi), 1 per-site pi        */
/* unrounded, 2 raw pi*W (mean pairwise differences; textbook).                */
/* s_scope: 0 S over all rows (run_tajd.sh:126,148), 1 S within subset P.      */
ORACLE_API int oracle_window_allpairs(const uint64_t *bits, uint64_t stride, uint32_t n, uint64_t s0, uint64_t s1,
                                      const uint64_t *mp, const uint64_t *ma, const uint64_t *mb,
                                      double seq_len, int d_pi_mode, int s_scope,
                                      uint32_t *ints, uint64_t *sums, double *dbl) {
    uint64_t W = s1 - s0;
    uint32_t mw = (n + 63) / 64;
    uint64_t *ea = (uint64_t *)malloc((size_t)(mw ? mw : 1) * 8), *eb = (uint64_t *)malloc((size_t)(mw ? mw : 1) * 8);
    for (uint32_t k = 0; k < mw; ++k) { ea[k] = ma[k] & ~mb[k]; eb[k] = mb[k] & ~ma[k]; } /* h-fst.py:181-185 */
    ma = ea; mb = eb;
    int64_t *I = (int64_t *)malloc((size_t)n * n * sizeof(int64_t));
    double *sim = (double *)malloc((size_t)n * n * sizeof(double));
    oracle_pairwise_counts(bits, stride, n, s0, s1, I);
    oracle_identity(I, n, W, 0, sim);
    oracle_site_scan(bits, stride, n, s0, s1, mp, ma, mb, ints, sums);
    /* subset P as a dense sub-matrix in index order (names sorted) */
    uint32_t nP = 0; uint32_t *idx = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) if (mask_has(mp, i)) idx[nP++] = i;
    double *sub = (double *)malloc((size_t)(nP ? nP : 1) * (nP ? nP : 1) * sizeof(double));
    for (uint32_t a = 0; a < nP; ++a) for (uint32_t b = 0; b < nP; ++b) sub[(size_t)a * nP + b] = sim[(size_t)idx[a] * n + idx[b]];
    double pi = 0, pi_site = 0;
    oracle_pica2(sub, nP, 1.0, -1, seq_len, &pi, &pi_site, NULL, NULL);
    uint8_t *fa = (uint8_t *)malloc(n ? n : 1), *fb = (uint8_t *)malloc(n ? n : 1);
    for (uint32_t i = 0; i < n; ++i) { fa[i] = (uint8_t)mask_has(ma, i); fb[i] = (uint8_t)mask_has(mb, i); }
    double f[6]; oracle_hfst(sim, n, fa, fb, seq_len, -1, f, NULL);
    dbl[0] = pi; dbl[1] = pi_site; dbl[2] = f[1]; dbl[3] = f[2]; dbl[4] = f[3]; dbl[5] = f[4]; dbl[6] = f[5]; dbl[7] = f[0];
    double S = (double)(s_scope == 0 ? ints[1] : ints[2]);
    double pin;
    if (d_pi_mode == 0) pin = oracle_py_round(pi_site, 8);     /* "%.8f" -> awk -> float(): run_tajd.sh:174,180 */
    else if (d_pi_mode == 1) pin = pi_site;
    else pin = pi * (double)W;
    double D = NAN;
    if (nP >= 2 && !isnan(pin)) oracle_tajimas_d(nP, S, pin, &D, NULL);
    dbl[8] = D;
    free(I); free(sim); free(idx); free(sub); free(fa); free(fb); free(ea); free(eb);
    return 0;
}

/* Same record from the site-count identities of SURVEY.md A.1 (the formulation */
/* the HIP scan kernel uses): sum_{i<j} H_ij = sum_s c_s (n - c_s) etc.  Used   */
/* to cross-check the formulation on the CPU and as the "fast port" CPU timing. */
/* Operates on 64-bit words of a site-major copy for speed.                    */
ORACLE_API int oracle_window_sitecount(const uint64_t *bits, uint64_t stride, uint32_t n, uint64_t s0, uint64_t s1,
                                       const uint64_t *mp, const uint64_t *ma, const uint64_t *mb,
                                       double seq_len, int d_pi_mode, int s_scope,
                                       uint32_t *ints, uint64_t *sums, double *dbl) {
    uint32_t mw = (n + 63) / 64;
    uint64_t *ea = (uint64_t *)malloc((size_t)(mw ? mw : 1) * 8), *eb = (uint64_t *)malloc((size_t)(mw ? mw : 1) * 8);
    for (uint32_t k = 0; k < mw; ++k) { ea[k] = ma[k] & ~mb[k]; eb[k] = mb[k] & ~ma[k]; } /* h-fst.py:181-185 */
    oracle_site_scan(bits, stride, n, s0, s1, mp, ea, eb, ints, sums);
    uint32_t nP = 0, nA = 0, nB = 0;
    for (uint32_t i = 0; i < n; ++i) { nP += mask_has(mp, i); nA += mask_has(ea, i); nB += mask_has(eb, i); }
    free(ea); free(eb);
    double W = (double)(s1 - s0);
    double pairsP = (double)nP * (double)(nP - 1) / 2.0;
    double pi = (nP >= 2 && W > 0) ? (double)sums[0] / (pairsP * W) : 0.0;
    double pi_site = (seq_len != 0.0) ? pi / seq_len : NAN;
    double pairsA = (double)nA * (double)(nA - 1.0) / 2.0, pairsB = (double)nB * (double)(nB - 1.0) / 2.0;
    double pi_a = (nA >= 2 && W > 0) ? (double)sums[1] / (pairsA * W) : 0.0;
    double pi_b = (nB >= 2 && W > 0) ? (double)sums[2] / (pairsB * W) : 0.0;
    double dxy = (nA && nB && W > 0) ? (double)sums[3] / ((double)nA * (double)nB * W) : 0.0;
    double pi_xy = 0.5 * (pi_a + pi_b);
    double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;
    double da = dxy - pi_xy;
    if (seq_len > 0) { pi_a /= seq_len; pi_b /= seq_len; da = (dxy - pi_xy) / seq_len; pi_xy /= seq_len; dxy /= seq_len; }
    dbl[0] = pi; dbl[1] = pi_site; dbl[2] = pi_a; dbl[3] = pi_b; dbl[4] = pi_xy; dbl[5] = dxy; dbl[6] = da; dbl[7] = fst;
    double S = (double)(s_scope == 0 ? ints[1] : ints[2]);
    double pin = d_pi_mode == 0 ? oracle_py_round(pi_site, 8) : d_pi_mode == 1 ? pi_site : pi * W;
    double D = NAN;
    if (nP >= 2 && !isnan(pin)) oracle_tajimas_d(nP, S, pin, &D, NULL);
    dbl[8] = D;
    return 0;
}

/* Fast site-count scan for CPU timing (bench.py cpu_baseline "sitecount_port"): */
/* site-major packed words (wps 64-bit words per site), same integer outputs.   */
ORACLE_API int oracle_site_scan_sitemajor(const uint64_t *sm, uint32_t wps64, uint32_t n, uint64_t s0, uint64_t s1,
                                          const uint64_t *mp, const uint64_t *ma, const uint64_t *mb,
                                          uint32_t *ints, uint64_t *sums) {
    uint32_t nP = 0, nA = 0, nB = 0;
    for (uint32_t i = 0; i < n; ++i) { nP += mask_has(mp, i); nA += mask_has(ma, i); nB += mask_has(mb, i); }
    uint32_t S_all = 0, S_P = 0, S_A = 0, S_B = 0;
    uint64_t sumP = 0, sumA = 0, sumB = 0, sumAB = 0;
    for (uint64_t s = s0; s < s1; ++s) {
        const uint64_t *w = sm + (size_t)s * wps64;
        uint32_t c = 0, cP = 0, cA = 0, cB = 0;
        for (uint32_t k = 0; k < wps64; ++k) {
            c += (uint32_t)__builtin_popcountll(w[k]);
            cP += (uint32_t)__builtin_popcountll(w[k] & mp[k]);
            cA += (uint32_t)__builtin_popcountll(w[k] & ma[k]);
            cB += (uint32_t)__builtin_popcountll(w[k] & mb[k]);
        }
        S_all += (c > 0 && c < n); S_P += (cP > 0 && cP < nP); S_A += (cA > 0 && cA < nA); S_B += (cB > 0 && cB < nB);
        sumP += (uint64_t)cP * (nP - cP); sumA += (uint64_t)cA * (nA - cA); sumB += (uint64_t)cB * (nB - cB);
        sumAB += (uint64_t)cA * (nB - cB) + (uint64_t)cB * (nA - cA);
    }
    ints[0] = (uint32_t)(s1 - s0); ints[1] = S_all; ints[2] = S_P; ints[3] = S_A; ints[4] = S_B;
    sums[0] = sumP; sums[1] = sumA; sums[2] = sumB; sums[3] = sumAB;
    return 0;
}

/* All-core form of the timing port: consecutive windows of `window_sites` sites over one
 * site-major slab, OpenMP over windows (bench.py cpu_baseline "sitecount_port_allcores"). */
ORACLE_API int oracle_site_scan_sitemajor_windows(const uint64_t *sm, uint32_t wps64, uint32_t n, uint64_t window_sites,
                                                  uint64_t n_win, const uint64_t *mp, const uint64_t *ma, const uint64_t *mb,
                                                  int threads, uint32_t *ints, uint64_t *sums) {
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t w = 0; w < (int64_t)n_win; ++w)
        oracle_site_scan_sitemajor(sm, wps64, n, (uint64_t)w * window_sites, (uint64_t)(w + 1) * window_sites, mp, ma, mb,
                                   ints + 8 * w, sums + 4 * w);
    return 0;
}

/* hap-major -> site-major (64-bit words per site) helper for the timing port. */
ORACLE_API int oracle_to_sitemajor(const uint64_t *bits, uint64_t stride, uint32_t n, uint64_t n_site, uint64_t *sm, uint32_t wps64) {
    memset(sm, 0, (size_t)n_site * wps64 * sizeof(uint64_t));
    for (uint32_t i = 0; i < n; ++i)
        for (uint64_t s = 0; s < n_site; ++s)
            if (bit_at(bits, stride, i, s)) sm[(size_t)s * wps64 + (i >> 6)] |= 1ull << (i & 63);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* scripts/hudson/hud.py "grouped" method (hud.py:64-128, 173-300).            */
/* group_sequences (:64-86): pica2-style greedy groups INSIDE a population     */
/* (seed = smallest remaining index here; the reference pops an arbitrary set  */
/* member); get_group_similarity (:88-99): the first pair (member of g1 in     */
/* sorted order x member of g2 in sorted order) present in the table.          */
static double hud_first_found(const double *sim, uint32_t n, const uint32_t *ia, const uint32_t *ga, uint32_t ma, uint32_t g1,
                              const uint32_t *ib, const uint32_t *gb, uint32_t mb, uint32_t g2, int rd) {
    for (uint32_t i = 0; i < ma; ++i) { if (ga[i] != g1) continue;
        for (uint32_t j = 0; j < mb; ++j) { if (gb[j] != g2) continue;
            uint32_t a = ia[i], b = ib[j];
            double v = a <= b ? sim[(size_t)a * n + b] : sim[(size_t)b * n + a];
            if (isnan(v)) continue;
            return rd >= 0 ? oracle_py_round(v, rd) : v;  /* :94-97 */
        } }
    return NAN;
}
static double hud_pi_grouped(const double *sim, uint32_t n, const uint32_t *idx, uint32_t m, double thr, int rd,
                             const uint32_t *seed_rank, uint32_t *grp, uint32_t *G_out, uint64_t *missing) {
    uint32_t G = greedy_groups(sim, n, idx, m, thr, rd, seed_rank, grp, NULL); /* hud.py:64-86 */
    *G_out = G; *missing = 0;
    if (m <= 1) return 0.0;                               /* :106-107 */
    uint32_t *sz = (uint32_t *)calloc(G ? G : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < m; ++i) sz[grp[i]]++;
    double acc = 0.0;
    for (uint32_t i = 0; i < G; ++i)
        for (uint32_t j = i + 1; j < G; ++j) {            /* :114-115 */
            double s = hud_first_found(sim, n, idx, grp, m, i, idx, grp, m, j, rd);
            if (isnan(s)) { ++*missing; continue; }
            double fi = (double)sz[i] / (double)m, fj = (double)sz[j] / (double)m;
            acc += 2 * fi * fj * (1 - s);                 /* :121 */
        }
    free(sz);
    return acc * (double)m / (double)(m - 1);             /* :127 */
}
/* out[6] = fst, pi_a, pi_b, pi_xy, dxy, da; counts[6] = groups_a, miss_a, groups_b, miss_b, group pairs, miss_between */
ORACLE_API int oracle_hud_grouped_seeded(const double *sim, uint32_t n, const uint8_t *in_a, const uint8_t *in_b, double threshold,
                                         int round_digits, double seq_len, const uint32_t *seed_rank, double *out, uint64_t *counts) {
    uint32_t *ia = (uint32_t *)malloc((n ? n : 1) * 4), *ib = (uint32_t *)malloc((n ? n : 1) * 4);
    uint32_t *ga = (uint32_t *)malloc((n ? n : 1) * 4), *gb = (uint32_t *)malloc((n ? n : 1) * 4);
    uint32_t ma = 0, mb = 0;
    for (uint32_t i = 0; i < n; ++i) {                     /* overlap removed from both: hud.py:186-190 */
        int ov = in_a[i] && in_b[i];
        if (in_a[i] && !ov) ia[ma++] = i;
        if (in_b[i] && !ov) ib[mb++] = i;
    }
    uint32_t GA, GB; uint64_t missA, missB, missX = 0, pairsX = 0;
    double pi_a = hud_pi_grouped(sim, n, ia, ma, threshold, round_digits, seed_rank, ga, &GA, &missA);
    double pi_b = hud_pi_grouped(sim, n, ib, mb, threshold, round_digits, seed_rank, gb, &GB, &missB);
    double pi_xy = 0.5 * (pi_a + pi_b);
    uint32_t *sa = (uint32_t *)calloc(GA ? GA : 1, 4), *sb = (uint32_t *)calloc(GB ? GB : 1, 4);
    for (uint32_t i = 0; i < ma; ++i) sa[ga[i]]++;
    for (uint32_t i = 0; i < mb; ++i) sb[gb[i]]++;
    double dxy = 0.0;
    for (uint32_t x = 0; x < GA; ++x)
        for (uint32_t y = 0; y < GB; ++y) {                /* hud.py:247-259 */
            double s = hud_first_found(sim, n, ia, ga, ma, x, ib, gb, mb, y, round_digits);
            if (isnan(s)) { ++missX; continue; }
            double w = ((double)sa[x] * (double)sb[y]) / ((double)ma * (double)mb);
            dxy += w * (1 - s);
            ++pairsX;
        }
    double fst = (dxy > 0) ? (dxy - pi_xy) / dxy : 0.0;
    if (seq_len > 0) { out[0] = fst; out[1] = pi_a / seq_len; out[2] = pi_b / seq_len; out[3] = pi_xy / seq_len; out[4] = dxy / seq_len; out[5] = (dxy - pi_xy) / seq_len; }
    else { out[0] = fst; out[1] = pi_a; out[2] = pi_b; out[3] = pi_xy; out[4] = dxy; out[5] = dxy - pi_xy; }
    if (counts) { counts[0] = GA; counts[1] = missA; counts[2] = GB; counts[3] = missB; counts[4] = pairsX; counts[5] = missX; }
    free(ia); free(ib); free(ga); free(gb); free(sa); free(sb);
    return 0;
}
ORACLE_API int oracle_hud_grouped(const double *sim, uint32_t n, const uint8_t *in_a, const uint8_t *in_b, double threshold,
                                  int round_digits, double seq_len, double *out, uint64_t *counts) {
    return oracle_hud_grouped_seeded(sim, n, in_a, in_b, threshold, round_digits, seq_len, NULL, out, counts);
}

/* ---- EHH: calc_EHH of scripts/wip/ehhgfa.py:6-21 (and ehh2.py:76-89) ----------------------------
 * Literal restatement on the packed matrix: for every prefix length i+1 count the pairs of member
 * rows that are equal on all of window sites 0..i (kept incrementally in `alive`: a pair that
 * differed once never matches a longer prefix), divide by m(m-1)/2, CPython round(, 3).
 * reverse != 0 = calc_EHH of the column-flipped window (ehhgfa.py:60-61).  m < 2 -> 500. */
ORACLE_API void oracle_ehh(const uint64_t *bits, uint64_t words, uint32_t n, uint64_t s0, uint64_t s1, const uint8_t *member,
                int reverse, double *out) {
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (!member || member[i]) idx[m++] = i;
    const uint64_t W = s1 - s0;
    if (m < 2) {
        for (uint64_t i = 0; i < W; ++i) out[i] = 500.0;
        free(idx);
        return;
    }
    const uint64_t n_pairs = (uint64_t)m * (m - 1) / 2;
    uint8_t *alive = (uint8_t *)malloc(n_pairs);
    memset(alive, 1, n_pairs);
    const double denom = (double)((uint64_t)m * (m - 1)) / 2.0;
    for (uint64_t i = 0; i < W; ++i) {
        const uint64_t s = reverse ? s1 - 1 - i : s0 + i;
        uint64_t homozygous = 0, p = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const uint64_t bj = (bits[(uint64_t)idx[j] * words + (s >> 6)] >> (s & 63)) & 1u;
            for (uint32_t k = j + 1; k < m; ++k, ++p) {
                const uint64_t bk = (bits[(uint64_t)idx[k] * words + (s >> 6)] >> (s & 63)) & 1u;
                if (bj != bk) alive[p] = 0;
                homozygous += alive[p];
            }
        }
        out[i] = oracle_py_round((double)homozygous / denom, 3);
    }
    free(alive);
    free(idx);
}
