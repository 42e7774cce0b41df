#!/usr/bin/env python3
"""All-pairs path on sliding windows: elementary segments shared between overlapping windows are contracted
once (impop_pairwise_scan).  Steady-state wall time per call for several window / step shapes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd

n = 465
ctx = impop_amd.Context(0)
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
out = {}
for size, step, NW in ((10000, 5000, 4096), (50000, 25000, 2048), (50000, 10000, 2048), (50000, 50000, 2048)):
    W = step * (NW + size // step)
    bm = ctx.synthetic(n, W, seed=4, keep_hap_major=True)
    wins = impop_amd.fixed_windows(W, size, step)[:NW]
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        bm.pairwise_scan(wins, None, in_a, in_b, threshold=0.999, round_digits=5)
        best = min(best, time.perf_counter() - t0)
    out[f"{size}_step_{step}"] = {"windows": len(wins), "ms": best * 1e3, "windows_per_s": len(wins) / best,
                                   "us_per_window": best / len(wins) * 1e6}
    bm.free()
print(json.dumps(out))
