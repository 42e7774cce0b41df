#!/usr/bin/env python3
"""Node-level matrix shaped like a pangenome graph: long anchor nodes every haplotype walks through (most of the base
pairs, the high bits of the weights) between short variant nodes.  All-pairs path on the weighted matrix and on the same
matrix compacted to its variable nodes (impop_matrix_compact of a weighted source: the anchors become a per-window
constant): same records, the contraction loses the anchors AND their weight planes.  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd

n, NW, per = 465, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2656
rng = np.random.default_rng(11)
K = NW * per
is_anchor = rng.random(K) < 0.4
nf = 12
founders = (rng.random((nf, K)) < 0.5).astype(np.uint8)
who = rng.integers(0, nf, size=n)
m = founders[who]
m ^= (rng.random((n, K)) < 0.0005).astype(np.uint8)
m[:, is_anchor] = 1
length = np.where(is_anchor, rng.integers(20, 3000, size=K), np.minimum(rng.geometric(0.3, size=K), 60)).astype(np.uint32)
ctx = impop_amd.Context(0)
bm = ctx.upload_dense(m, keep_hap_major=True)
del m, founders
bm.set_site_weights(length)
cum = np.concatenate(([0], np.cumsum(length.astype(np.int64))))
wins = [(k * per, (k + 1) * per, int(cum[(k + 1) * per] - cum[k * per])) for k in range(NW)]
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
t0 = time.perf_counter()
cm = bm.compact()
t_compact = time.perf_counter() - t0
out = {"n_hap": n, "windows": NW, "nodes_per_window": per, "mean_bp_per_window": float(cum[-1]) / NW, "anchor_fraction_of_nodes": 0.4,
       "kept_nodes": cm.n_site, "of_nodes": K, "compact_seconds": t_compact}
res = {}
for name, mat in (("weighted", bm), ("weighted_compacted", cm)):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = mat.pairwise_scan(wins, None, in_a, in_b, threshold=0.999, round_digits=5, s_scope=2)
        best = min(best, time.perf_counter() - t0)
    res[name] = r
    out[name] = {"s_per_batch": best, "windows_per_s": NW / best, "mean_groups": float(r["n_groups"].mean())}
out["records_identical"] = res["weighted"].tobytes() == res["weighted_compacted"].tobytes()
print(json.dumps(out))
