#!/usr/bin/env python3
"""BASELINE config 5 at full size on the all-pairs path: 4096 haplotypes x 10^7 sites resident (SB64 + RB32 = 10.3 GB),
(i) ONE window, K-split FP4 Gram (impop_pairwise_counts), (ii) 200 x 50 000-site windows (Gram only and the full
impop_pairwise_scan), plus the streaming scan of the same shapes.  Wall-clock around the C-ABI calls (copies
included, stated per row); run under `rocprofv3 --kernel-trace --stats` for kernel-only durations.

    python tools/bench_config5.py [--out profiles/rNN_config5.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FP4_DENSE_PEAK_MACS = 5.0e15  # MI355X_MICROARCH.md: ~10 PFLOP/s dense FP4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--n-hap", type=int, default=4096)
    ap.add_argument("--sites", type=int, default=10_000_000)
    args = ap.parse_args()
    import numpy as np

    import impop_amd
    n, W = args.n_hap, args.sites
    ctx = impop_amd.Context(0)
    t0 = time.perf_counter()
    bm = ctx.synthetic(n, W, seed=5, n_founder=16, p_founder=0.05, p_private_word=0.05, keep_hap_major=True)
    ctx.synchronize()
    out = {"n_hap": n, "n_site": W, "device_bytes": bm.device_bytes, "synth_s": time.perf_counter() - t0, "rows": []}
    pair_macs = n * (n + 1) // 2

    def timed(label, fn, macs, reps=3, note=""):
        fn()  # warm-up (scratch, code objects, cached site bitmap)
        ctx.synchronize()
        best = 1e30
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t)
        row = {"what": label, "ms": best * 1e3, "note": note}
        if macs:
            row.update({"algorithmic_macs": macs, "macs_per_s": macs / best, "frac_of_fp4_dense_peak": macs / best / FP4_DENSE_PEAK_MACS})
        out["rows"].append(row)
        print(json.dumps(row), flush=True)

    timed("gram_single_window", lambda: bm.pairwise_counts(0, W), pair_macs * W,
          note="impop_pairwise_counts: K-split Gram + symmetrise + 67 MB copy to the host")
    wins200 = impop_amd.fixed_windows(200 * 50000, 50000)
    in_a = np.arange(n) < 1000
    in_b = np.arange(n) >= 3000
    timed("pairwise_scan_200x50kb", lambda: bm.pairwise_scan(wins200, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5),
          pair_macs * 50000 * 200, reps=2, note="Gram + pica2 (-t 0.999 -r 5) + h-fst + S + D for 200 windows, chunks of <= 128 Gram matrices")
    timed("pairwise_scan_single_window", lambda: bm.pairwise_scan([(0, W, W)], None, in_a, in_b, kind="match", threshold=0.999, round_digits=5),
          pair_macs * W, reps=2, note="one 10^7-site window end to end")
    for label, wins in (("scan_single_window", [(0, W, W)]), ("scan_200x50kb", wins200)):
        plan = bm.plan(wins, None, in_a, in_b)
        plan.launch(); ctx.synchronize()
        plan.timing(True)
        t = time.perf_counter()
        for _ in range(10):
            plan.launch()
        ms, k = plan.elapsed()
        wall = (time.perf_counter() - t) / 10
        row = {"what": label, "kernel_ms": ms / k, "step_ms_wall": wall * 1e3, "tiles": plan.n_tiles,
               "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9, "frac_of_8TBps": plan.bytes_streamed / (ms / k / 1e3) / 8e12}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
        plan.destroy()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    bm.free()
    ctx.close()


if __name__ == "__main__":
    main()
