#!/usr/bin/env python3
"""Host-side phase times of impop_pairwise_scan (IMPOP_TRACE=1) on the bench shape, full and compacted matrix; plus the Python
marshalling around the C call.  Run on the GPU box: IMPOP_TRACE=1 python tools/trace_pairwise_host.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import impop_amd

ctx = impop_amd.Context(0)
n, W, NW = 465, 50000, 4096
bm = ctx.synthetic(n, W * NW, keep_hap_major=True)
wins = impop_amd.fixed_windows(W * NW, W)
in_a = np.zeros(n, np.uint8); in_a[:140] = 1
in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
kw = dict(kind="match", threshold=0.999, round_digits=5)
for name, mat in (("full", bm), ("compact", bm.compact())):
    mat.pairwise_scan(wins, None, in_a, in_b, **kw)
    for rep in range(2):
        print(f"--- {name} call {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        mat.pairwise_scan(wins, None, in_a, in_b, **kw)
        print(f"--- python-side total {1e6 * (time.perf_counter() - t0):.1f} us", file=sys.stderr, flush=True)
