#!/usr/bin/env python3
"""Soak test (not part of the suite): pica2's grouping on bits for LARGE problems (>= 1024 elements: pica2_adj_kernel +
greedy_groups_bits<true>, Step 2 split over workgroups) and mid-size ones (in-kernel blocks), group by group against the
oracle — Gram problems straight from the bit matrix (with and without a population subset) and dense tables with
missing pairs."""
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
while time.time() < t_end:
    n = int(rng.choice([65, 255, 256, 257, 511, 513, 1023, 1024, 1025, 1500, 2048, 2049, 3000, 4096, 4100]))
    W = int(rng.integers(100, 1500))
    nf = int(rng.integers(1, 40))
    f = (rng.random((nf, W)) < 0.5).astype(np.uint8)
    m = f[rng.integers(0, nf, size=n)] ^ (rng.random((n, W)) < rng.choice([0.0, 0.0005, 0.003, 0.02])).astype(np.uint8)
    bits = orc.pack_hap_major(m)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    a = int(rng.integers(0, W // 2)); b = int(rng.integers(a + 1, W + 1))
    kind = "match" if rng.random() < 0.7 else "dice"
    sim = orc.identity(orc.pairwise_counts(bits, n, a, b), b - a, 0 if kind == "match" else 1)
    thr = float(rng.choice([1.0, 0.9999, 0.999, 0.995, 0.99, 0.9]))
    rd = None if rng.random() < 0.5 else int(rng.integers(2, 6))
    inP = None if rng.random() < 0.5 else (rng.random(n) < 0.9).astype(np.uint8)
    r = bm.pairwise_scan([(a, b, b - a)], inP, None, None, kind=kind, threshold=thr, round_digits=rd, s_scope=2)[0]
    sel = np.arange(n) if inP is None else np.nonzero(inP)[0]
    pi, ps, grp, G = orc.pica2(sim[np.ix_(sel, sel)], thr, b - a, rd)
    assert int(r["n_groups"]) == G, (n, W, a, b, kind, thr, rd, int(r["n_groups"]), G)
    assert abs(float(r["pi"]) - pi) <= 1e-9 * max(abs(pi), 1e-300), (n, W, a, b, kind, thr, rd, float(r["pi"]), pi)
    # the dense-table entry, with missing pairs, element by element
    if rng.random() < 0.5:
        drop = np.triu(rng.random((n, n)) < 0.005, 1)
        sim[drop | drop.T] = np.nan
    gpi, gps, ggrp, gG = ctx.pi_from_identity(sim, thr, rd, b - a)
    opi, ops, ogrp, oG = orc.pica2(sim, thr, b - a, rd)
    assert gG == oG and (ggrp == ogrp).all(), (n, W, "dense", thr, rd, gG, oG)
    assert abs(gpi - opi) <= 1e-9 * max(abs(opi), 1e-300), (n, W, "dense pi", gpi, opi)
    bm.free()
    it += 1
    if it % 5 == 0:
        print("iterations", it, flush=True)
print("soak ok:", it, "iterations")
