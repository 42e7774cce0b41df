#!/usr/bin/env python3
"""K-population scan (impop_scan_multi) on the chr2-scale workload: wall time per call, K = 2..8."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import impop_amd
n, W, NW = 465, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4854
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, W * NW, seed=20251031)
wins = impop_amd.fixed_windows(W * NW, W)
out = {}
for K in (2, 5, 8):
    sizes = [n // K] * K
    pops, o = [], 0
    for s in sizes:
        f = np.zeros(n, np.uint8); f[o:o + s] = 1; o += s
        pops.append(f)
    bm.scan_multi(wins[:8], pops)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); bm.scan_multi(wins, pops); best = min(best, time.perf_counter() - t0)
    out[f"K{K}"] = {"ms": best * 1e3, "windows_per_s": NW / best, "layout_GBps_incl_overheads": bm.device_bytes / best / 1e9}
print(json.dumps(out))
