#!/usr/bin/env python3
"""Soak test (not part of the suite): random matrices / windows / masks through the streaming scan, the
compacted-matrix scan, the K-population scan, the allele-frequency spectrum and per-site counts, checked against the
oracle's site-count formulation / numpy."""
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
INT = ("n_sites", "s_all", "s_p", "s_a", "s_b", "sum_p", "sum_a", "sum_b", "sum_ab")
DBL = ("pi", "pi_site", "pi_a", "pi_b", "pi_xy", "dxy", "da", "fst", "tajima_d")
while time.time() < t_end:
    n = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 65, 200, 465, 512, 513, 700, 1100]))
    W = int(rng.integers(1, 30000))
    m = (rng.random((n, W)) < rng.random() * rng.random(W)).astype(np.uint8)
    bits = orc.pack_hap_major(m)
    bm = ctx.upload_dense(m, keep_hap_major=False)
    wins = []
    for _ in range(int(rng.integers(1, 20))):
        a = int(rng.integers(0, W)); b = int(rng.integers(a, W + 1))
        wins.append((a, b, int(rng.integers(0, 3)) * max(b - a, 1)))
    inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
    inP = None if rng.random() < 0.5 else (rng.random(n) < 0.7).astype(np.uint8)
    mode, scope = int(rng.integers(0, 3)), int(rng.integers(0, 2))
    tb = int(rng.choice([0, 1, 3, 32, 100]))
    got = bm.scan(wins, inP, inA, inB, d_pi_mode=mode, s_scope=scope, tile_blocks=tb)
    cm = bm.compact()
    assert cm.scan(wins, inP, inA, inB, d_pi_mode=mode, s_scope=scope).tobytes() == got.tobytes(), (n, W, "compact")
    cm.free()
    mp = orc.pack_mask(np.ones(n, np.uint8) if inP is None else inP)
    for (a, b, L), r in zip(wins[:6], got):
        want = orc.window_sitecount(bits, n, a, b, mp, orc.pack_mask(inA), orc.pack_mask(inB), L, mode, scope)
        for k in INT:
            assert int(r[k]) == int(want[k]), (n, W, (a, b), k)
        for k in DBL:
            x, y = float(r[k]), float(want[k])
            assert (x != x and y != y) or abs(x - y) <= 1e-9 * max(abs(x), abs(y)), (n, W, (a, b), k, x, y)
    if n >= 3 and n <= 65535:
        K = int(rng.integers(2, min(8, n) + 1))
        lab = rng.integers(0, K + 1, size=n)
        pops = [(lab == k).astype(np.uint8) for k in range(K)]
        pr = bm.scan_multi(wins, pops)
        p = 0
        for i in range(K):
            for j in range(i + 1, K):
                two = bm.scan(wins, None, pops[i], pops[j])
                for k in ("fst", "pi_a", "pi_b", "pi_xy", "dxy", "da"):
                    x, y = pr[:, p][k], two[k]
                    assert ((x == y) | (np.isnan(x) & np.isnan(y))).all(), (n, W, i, j, k)
                p += 1
    # allele-frequency spectrum and per-site counts of the same windows against numpy (few windows: several workgroups
    # per window; every few iterations thousands of short windows: one workgroup per window)
    rows = np.ones(n, bool) if inP is None else inP.astype(bool)
    cs = m[rows].sum(0).astype(np.int64) if rows.any() else np.zeros(W, np.int64)
    aw = [(a, b) for a, b, _ in wins]
    if it % 5 == 0 and W > 80:
        st = np.sort(rng.integers(0, W - 70, size=8300))
        aw = [(int(a), int(a + rng.integers(0, 70))) for a in st]
    spec = bm.afs(aw, inP)
    for (a, b), g in list(zip(aw, spec))[:: max(1, len(aw) // 40)]:
        assert (g == np.bincount(cs[a:b], minlength=int(rows.sum()) + 1)).all(), (n, W, a, b, "afs")
    assert (spec.sum(axis=1) == np.array([b - a for a, b in aw])).all(), (n, W, "afs totals")
    a, b = wins[0][0], wins[0][1]
    if b > a:
        assert (bm.site_counts(a, b, inP).astype(np.int64) == cs[a:b]).all(), (n, W, a, b, "site counts")
    bm.free()
    it += 1
    if it % 20 == 0:
        print("iterations", it, flush=True)
print("soak ok:", it, "iterations")
