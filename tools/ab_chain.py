#!/usr/bin/env python3
"""Repeated wall times of impop_pairwise_scan on short-window shapes (compacted bench matrix; weighted node-level matrix; 10 kb / 5 kb
sliding windows) — run with IMPOP_GRAM_CHAIN=1 (no chains) and unset (auto) to compare."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import impop_amd
ctx = impop_amd.Context(0)
n, W, NW = 465, 50000, 4096
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
kw = dict(kind="match", threshold=0.999, round_digits=5)
def best(f, reps=7):
    f(); ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3
bm = ctx.synthetic(n, W * NW, keep_hap_major=True)
wins = impop_amd.fixed_windows(W * NW, W)
cm = bm.compact()
print("compacted 4096 x 50kb: best %.3f median %.3f ms" % best(lambda: cm.pairwise_scan(wins, None, in_a, in_b, **kw)), flush=True)
sw = impop_amd.fixed_windows(10000 * 2049, 10000, 5000)[:4096]
print("sliding 10kb/5kb x 4096: best %.3f median %.3f ms" % best(lambda: bm.pairwise_scan(sw, None, in_a, in_b, **kw)), flush=True)
cm.free(); bm.free()
# node-level weighted: 2656 nodes per window, lengths up to 2999
rng = np.random.default_rng(1)
nodes = 2656
nm = ctx.synthetic(n, nodes * NW, seed=3, n_founder=8, p_founder=0.02, p_private_word=0.05, keep_hap_major=True)
nm.set_site_weights(rng.integers(1, 3000, size=nodes * NW).astype(np.uint32) % np.uint32(50) + 1)
nw = impop_amd.fixed_windows(nodes * NW, nodes)
print("weighted nodes (6 planes) 4096: best %.3f median %.3f ms" % best(lambda: nm.pairwise_scan(nw, None, in_a, in_b, s_scope=2, **kw)), flush=True)
