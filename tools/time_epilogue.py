#!/usr/bin/env python3
"""Wall time of impop_pairwise_scan on 4096 x (465 hap x 10 kb) windows (epilogue-heavy shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import impop_amd
n, W, NW = 465, 10000, 4096
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, W * NW, seed=20251031, keep_hap_major=True)
wins = impop_amd.fixed_windows(W * NW, W)
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
for thr in (0.999, 0.9999, 1.0):  # few groups ... (nearly) every haplotype its own group
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = bm.pairwise_scan(wins, None, in_a, in_b, threshold=thr, round_digits=5 if thr < 1.0 else None, s_scope=2)
        best = min(best, time.perf_counter() - t0)
    print(os.environ.get("IMPOP_HIP_LIBRARY", "default"), "threshold", thr, "mean groups", round(float(r["n_groups"].mean()), 1),
          "ms", round(best * 1e3, 3), "us/window", round(best / NW * 1e6, 3))
