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
    n = int(rng.choice([1, 2, 31, 32, 33, 95, 96, 97, 192, 193, 300, 465, 700, 1030, 1500]))
    W = int(rng.integers(1, 40000 if n < 1000 else 6000))
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
    inP = None if rng.random() < 0.6 else (rng.random(n) < 0.8).astype(np.uint8)
    meth = "grouped" if rng.random() < 0.3 else "direct"
    got = bm.pairwise_scan(wins, inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)
    for k, wdw in enumerate(wins):
        one = bm.pairwise_scan([wdw], inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth)[0]
        assert got[k].tobytes() == one.tobytes(), (n, W, wdw, thr, rd, kind, meth)
    # the same from the matrix compacted to its variable sites: byte-identical records, counts and identities
    cm = bm.compact()
    assert cm.pairwise_scan(wins, inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth).tobytes() == got.tobytes(), (n, W, "compact")
    if wins:
        a, b, _ = wins[0]
        assert (cm.pairwise_counts(a, b) == bm.pairwise_counts(a, b)).all()
        assert cm.pairwise_identity(a, b, kind).tobytes() == bm.pairwise_identity(a, b, kind).tobytes()
    cm.free()
    # node lengths as site weights == the bp-expanded matrix (small cases)
    if n <= 200 and W <= 3000:
        length = rng.integers(1, 50, size=W).astype(np.uint32)
        cum = np.concatenate(([0], np.cumsum(length))).astype(np.int64)
        bm.set_site_weights(length)
        be = ctx.upload_dense(np.repeat(m, length, axis=1), keep_hap_major=True)
        we = [(int(cum[a]), int(cum[b]), L) for a, b, L in wins]
        gw = bm.pairwise_scan(wins, inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth, s_scope=2)
        ge = be.pairwise_scan(we, inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth, s_scope=2)
        assert gw.tobytes() == ge.tobytes(), (n, W, "weighted")
        cw = bm.compact()  # weighted source: dropped all-ones columns come back as their summed weights
        assert cw.pairwise_scan(wins, inP, inA, inB, kind=kind, threshold=thr, round_digits=rd, fst_method=meth, s_scope=2).tobytes() == gw.tobytes(), (n, W, "weighted compact")
        cw.free()
        be.free()
        bm.set_site_weights(None)
    if n <= 200 and wins:
        a, b, L = wins[0]
        if b > a:
            sim = orc.identity(orc.pairwise_counts(orc.pack_hap_major(m), n, a, b), b - a, 0 if kind == "match" else 1)
            if inP is not None:
                sel = np.nonzero(inP)[0]
                simp = sim[np.ix_(sel, sel)]
            else:
                simp = sim
            pi, ps, _, G = orc.pica2(simp, thr, L if L else None, rd)
            assert abs(float(got[0]["pi"]) - pi) <= 1e-9 * max(abs(pi), 1e-300) and int(got[0]["n_groups"]) == G, (n, W, wins[0])
            if meth == "direct":
                h, _ = orc.hfst(sim, inA, inB, L if L else None, rd)
                for kk in ("fst", "pi_a", "pi_b", "dxy"):
                    x, y = float(got[0][kk]), h[kk]
                    # Fst is a difference of nearly equal means on unstructured matrices: absolute tolerance there
                    assert abs(x - y) <= (1e-9 if kk == "fst" else 1e-9 * max(abs(y), 1e-300)), (n, W, kk, x, y)
    bm.free()
    it += 1
    if it % 20 == 0:
        print("iterations", it, flush=True)
print("soak ok:", it, "iterations")
