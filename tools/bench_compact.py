#!/usr/bin/env python3
"""Secondary point of SURVEY §8(d): the headline workload (chr2-scale, 465 haplotypes, 50 kb windows)
scanned from a matrix compacted to its variable sites (impop_matrix_compact): same records, W/S times
fewer bytes.  Prints one JSON object; not the bench line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd

n, W, NW = 465, 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4854
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, W * NW, seed=20251031)
wins = impop_amd.fixed_windows(W * NW, W)
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
t0 = time.perf_counter()
cm = bm.compact()
ctx.synchronize()
t_compact = time.perf_counter() - t0
full = bm.plan(wins, None, in_a, in_b)
comp = cm.plan(wins, None, in_a, in_b)
res = {}
for name, plan in (("full", full), ("compact", comp)):
    plan.launch(); ctx.synchronize()
    plan.timing(True)
    for _ in range(20):
        plan.launch()
    ms, k = plan.elapsed()
    res[name] = {"kernel_ms": ms / k, "windows_per_s": NW / (ms / k / 1e3), "bytes_streamed": plan.bytes_streamed,
                 "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9}
a, b = full.fetch(), comp.fetch()
identical = a.tobytes() == b.tobytes()
full.destroy(); comp.destroy()
print(json.dumps({"n_hap": n, "windows": NW, "window_sites": W, "sites": W * NW, "variable_sites": cm.n_site,
                  "variable_per_window": cm.n_site / NW, "compact_seconds": t_compact,
                  "compact_GBps_of_input": bm.device_bytes / t_compact / 1e9, "records_identical": identical, **res}))
cm.free(); bm.free(); ctx.close()
