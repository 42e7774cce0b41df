#!/usr/bin/env python3
"""impop_pairwise_scan on 4096 x (465 hap x 10 kb) windows at a 5 kb step (shared segment matrices), default chain; for
`rocprofv3 --kernel-trace` (tools/prof_epilogue.sh style summaries) or plain wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import impop_amd
n = 465
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, 10000 * 2050, seed=20251031, keep_hap_major=True)
sw = impop_amd.fixed_windows(10000 * 2049, 10000, 5000)[:4096]
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
for thr in (0.999, 0.9999):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        r = bm.pairwise_scan(sw, None, in_a, in_b, threshold=thr, round_digits=5)
        best = min(best, time.perf_counter() - t0)
    print("sliding threshold", thr, "mean groups", round(float(r["n_groups"].mean()), 1), "ms", round(best * 1e3, 3))
