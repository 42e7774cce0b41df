#!/usr/bin/env python3
"""The shape real graphs arrive in: one column per graph NODE, node lengths as site weights (impop_gfa_parse ->
impop_matrix_set_site_weights).  465 haplotypes, 4096 windows of ~2650 nodes whose lengths add up to ~50 kb: the
scan path (weighted any-n kernel) and the all-pairs path (Gram over weight bit planes) against the same windows on
a bp-per-column matrix.  Prints one JSON object; not the bench line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd

n, NW, nodes_per_win = 465, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 2656
rng = np.random.default_rng(5)
n_node = NW * nodes_per_win
# node lengths: mostly 1 bp (SNV bubbles), a geometric body, a few long anchors; mean ~19 bp
length = np.ones(n_node, dtype=np.uint32)
body = rng.random(n_node) < 0.45
length[body] = rng.geometric(0.05, size=int(body.sum())).astype(np.uint32)
anchor = rng.random(n_node) < 0.01
length[anchor] = rng.integers(200, 3000, size=int(anchor.sum())).astype(np.uint32)
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, n_node, seed=20251031, keep_hap_major=True)
bm.set_site_weights(length)
cum = np.concatenate(([0], np.cumsum(length.astype(np.int64))))
wins = [(k * nodes_per_win, (k + 1) * nodes_per_win, int(cum[(k + 1) * nodes_per_win] - cum[k * nodes_per_win])) for k in range(NW)]
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
out = {"n_hap": n, "windows": NW, "nodes_per_window": nodes_per_win, "mean_bp_per_window": float(cum[-1]) / NW,
       "max_node_length": int(length.max()), "weight_planes": int(length.max()).bit_length()}
plan = bm.plan(wins, None, in_a, in_b)
plan.launch(); ctx.synchronize()
plan.timing(True)
for _ in range(10):
    plan.launch()
ms, k = plan.elapsed()
out["scan_weighted"] = {"kernel_ms": ms / k, "windows_per_s": NW / (ms / k / 1e3), "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9}
plan.destroy()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    r = bm.pairwise_scan(wins, None, in_a, in_b, threshold=0.999, round_digits=5, s_scope=2)
    best = min(best, time.perf_counter() - t0)
out["all_pairs_weighted"] = {"s_per_batch": best, "windows_per_s": NW / best, "mean_groups": float(r["n_groups"].mean())}
print(json.dumps(out))
bm.free(); ctx.close()
