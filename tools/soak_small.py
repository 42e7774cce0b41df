#!/usr/bin/env python3
"""Soak test (not part of the suite) of the window-statistics epilogue kernels (csrc/stats_small.hip): impop_pairwise_scan on
disjoint and sliding windows of 1 .. 512 haplotypes, `match` identity — subsets (element lists), overlapping / empty / one-member
populations, thresholds from "one group" to "nobody joins", roundings — every field against the oracle.
    python tools/soak_small.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = impop_amd.Context(0)
t_end = time.time() + budget
it = 0


def close(a, b, floor=0.0):
    if a != a or b != b:
        return a != a and b != b
    return abs(a - b) <= max(1e-9 * abs(b), floor)


while time.time() < t_end:
    n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 465, 511, 512]))
    nwin = int(rng.choice([1, 2, 5, 9]))
    wlen = int(rng.integers(40, 1200))
    step = wlen if rng.random() < 0.5 else max(1, wlen // int(rng.choice([2, 3, 5])))  # < wlen: sliding windows (segment sums)
    W = (nwin - 1) * step + wlen + int(rng.integers(0, 50))
    nf = int(rng.integers(1, 30))
    f = (rng.random((nf, W)) < 0.5).astype(np.uint8)
    m = f[rng.integers(0, nf, size=n)] ^ (rng.random((n, W)) < rng.choice([0.0, 0.0005, 0.003, 0.02])).astype(np.uint8)
    bits = orc.pack_hap_major(m)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    if rng.random() < 0.3:
        bm = bm.compact()
    wins = [(k * step, k * step + wlen, int(rng.choice([0, wlen, 50000]))) for k in range(nwin)]
    thr = float(rng.choice([1.5, 1.0, 0.9999, 0.999, 0.995, 0.99, 0.9, 0.5, 0.0, -1.0]))
    rd = None if rng.random() < 0.4 else int(rng.integers(0, 7))
    inP = None if rng.random() < 0.5 else (rng.random(n) < rng.choice([0.05, 0.5, 0.9])).astype(np.uint8)
    pa, pb = rng.choice([0.0, 0.02, 0.3, 0.5, 1.0], size=2)
    inA = (rng.random(n) < pa).astype(np.uint8)
    inB = (rng.random(n) < pb).astype(np.uint8)
    if rng.random() < 0.5:
        inB &= ~inA & 1  # disjoint, else overlapping (h-fst.py:181-185 drops the overlap)
    res = bm.pairwise_scan(wins, inP, inA, inB, kind="match", threshold=thr, round_digits=rd, s_scope=2)
    sel = np.arange(n) if inP is None else np.nonzero(inP)[0]
    for (a, b, L), r in zip(wins, res):
        sim = orc.identity(orc.pairwise_counts(bits, n, a, b), b - a, 0)
        pi, ps, grp, G = orc.pica2(sim[np.ix_(sel, sel)], thr, L if L else None, rd)
        ctxt = (n, W, (a, b, L), thr, rd, None if inP is None else int(inP.sum()), int(inA.sum()), int(inB.sum()))
        assert int(r["n_groups"]) == G, ctxt + (int(r["n_groups"]), G)
        assert close(float(r["pi"]), pi), ctxt + ("pi", float(r["pi"]), pi)
        assert close(float(r["pi_site"]), ps), ctxt + ("pi_site", float(r["pi_site"]), ps)
        h, _ = orc.hfst(sim, inA, inB, L if L else None, rd)
        for k, v in h.items():
            floor = 1e-12 if k == "fst" else 1e-12 * abs(h["dxy"]) if k == "da" else 0.0
            assert close(float(r[k]), v, floor), ctxt + (k, float(r[k]), v)
    bm.free()
    it += 1
    if it % 20 == 0:
        print("iterations", it, flush=True)
print("soak ok:", it, "iterations")
