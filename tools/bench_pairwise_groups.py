#!/usr/bin/env python3
"""All-pairs path when the pipeline's threshold sits INSIDE the distribution of pairwise distances (what -t 0.999 does on human data:
heterozygosity ~1e-3 = 50 differences per 50 kb window): 32 founders whose pairwise distances straddle the threshold, so a window
has tens to hundreds of groups.  Prints groups per window and the call's wall time; run under rocprofv3 --kernel-trace for the
kernels (tools/ab_gram.sh style)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import impop_amd

n, W, NW = 465, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = impop_amd.Context(0)
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
for nf, pf in ((8, 1e-3), (32, 5e-4), (32, 3e-4), (32, 2e-4)):
    bm = ctx.synthetic(n, W * NW, seed=7, n_founder=nf, p_founder=pf, keep_hap_major=True)
    wins = impop_amd.fixed_windows(W * NW, W)
    kw = dict(kind="match", threshold=0.999, round_digits=5)
    r = bm.pairwise_scan(wins, None, in_a, in_b, **kw)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = bm.pairwise_scan(wins, None, in_a, in_b, **kw)
        best = min(best, time.perf_counter() - t0)
    g = r["n_groups"]
    print(f"founders {nf} p_founder {pf}: groups per window mean {g.mean():.1f} min {g.min()} max {g.max()} | {best * 1e3:.2f} ms per {NW} windows = "
          f"{NW / best:.0f} windows/s", flush=True)
    bm.free()
