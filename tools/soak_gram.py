#!/usr/bin/env python3
"""Soak test (not part of the suite): random shapes through the Gram kernel and the epilogues, checked
against numpy / the oracle, for a given number of seconds."""
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
    n = int(rng.choice([1, 2, 31, 32, 33, 95, 96, 97, 192, 193, 300, 465, 700]))
    W = int(rng.integers(1, 40000))
    dens = rng.random()
    m = (rng.random((n, W)) < dens * rng.random(W)).astype(np.uint8)
    bm = ctx.upload_dense(m, keep_hap_major=True)
    mi = m.astype(np.int64)
    for _ in range(3):
        a = int(rng.integers(0, W)); b = int(rng.integers(a, W + 1))
        I = bm.pairwise_counts(a, b).astype(np.int64)
        w = mi[:, a:b]
        assert (I == w @ w.T).all(), (n, W, a, b)
    # windows batch (overlapping) through pairwise_scan vs one by one
    wins = []
    for _ in range(int(rng.integers(1, 12))):
        a = int(rng.integers(0, W)); b = int(rng.integers(a, min(W, a + 6000) + 1))
        wins.append((a, b, int(rng.integers(0, 3)) * (b - a)))
    inA = (rng.random(n) < 0.4).astype(np.uint8); inB = (rng.random(n) < 0.4).astype(np.uint8)
    thr = float(rng.choice([1.0, 0.999, 0.99, 0.9, 0.5]))
    rd = None if rng.random() < 0.5 else int(rng.integers(1, 6))
    kind = "match" if rng.random() < 0.6 else "dice"
    got = bm.pairwise_scan(wins, None, inA, inB, kind=kind, threshold=thr, round_digits=rd)
    for k, wdw in enumerate(wins):
        one = bm.pairwise_scan([wdw], None, inA, inB, kind=kind, threshold=thr, round_digits=rd)[0]
        assert got[k].tobytes() == one.tobytes(), (n, W, wdw, thr, rd, kind)
    if n <= 200 and wins:
        a, b, L = wins[0]
        if b > a:
            sim = orc.identity(orc.pairwise_counts(orc.pack_hap_major(m), n, a, b), b - a, 0 if kind == "match" else 1)
            pi, ps, _, G = orc.pica2(sim, thr, L if L else None, rd)
            assert abs(float(got[0]["pi"]) - pi) <= 1e-9 * max(abs(pi), 1e-300) and int(got[0]["n_groups"]) == G, (n, W, wins[0])
    bm.free()
    it += 1
    if it % 20 == 0:
        print("iterations", it, flush=True)
print("soak ok:", it, "iterations")
