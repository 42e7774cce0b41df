#!/usr/bin/env python3
"""Timing point for impop_ehh: n-hap synthetic founder matrix, one 50 kb window, both directions."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import impop_amd

ap = argparse.ArgumentParser()
ap.add_argument("--n-hap", type=int, default=465)
ap.add_argument("--window", type=int, default=50000)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
ctx = impop_amd.Context(0)
bm = ctx.synthetic(a.n_hap, a.window * 8, seed=20251031)
res = {}
for rev in (False, True):
    bm.ehh(0, a.window, reverse=rev)
    t0 = time.perf_counter()
    for r in range(a.reps):
        v = bm.ehh((r % 8) * a.window, (r % 8 + 1) * a.window, reverse=rev)
    dt = (time.perf_counter() - t0) / a.reps
    res["reverse" if rev else "forward"] = {"ms_per_window_vector": dt * 1e3, "ehh_first": float(v[0]), "ehh_last": float(v[-1])}
print(json.dumps({"n_hap": a.n_hap, "window_sites": a.window, **res}))
