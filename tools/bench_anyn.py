#!/usr/bin/env python3
"""Any-n scan kernel (n > 512 haplotypes) on the BASELINE config 5 shape, one process, several shapes / tile sizes:

    python tools/bench_anyn.py [--n-hap 4096] [--out profiles/rNN_anyn.json]

For each (windows x window length, tile_blocks): streaming-kernel time from the library's own HIP events
(impop_scan_plan_timing), layout GB/s and fraction of the 8 TB/s HBM peak; records are checked equal across tile
sizes.  Not part of the product path."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-hap", type=int, default=4096)
    ap.add_argument("--out", default=None)
    ap.add_argument("--launches", type=int, default=10)
    args = ap.parse_args()
    import numpy as np

    import impop_amd
    ctx = impop_amd.Context(0)
    n = args.n_hap
    in_a = np.zeros(n, np.uint8); in_a[: n // 3] = 1
    in_b = np.zeros(n, np.uint8); in_b[n // 3: n // 2] = 1
    rows = []
    for n_win, W in ((200, 50000), (1, 10_000_000), (1000, 50000)):
        bm = ctx.synthetic(n, n_win * W, seed=5)
        wins = impop_amd.fixed_windows(n_win * W, W)
        ref = None
        for tb in (0, 4, 8, 16, 32, 64):
            for masks, label in (((None, in_a, in_b), "P=all"), ((in_a | in_b, in_a, in_b), "P=subset")):
                if label == "P=subset" and tb != 0:
                    continue
                plan = bm.plan(wins, *masks, tile_blocks=tb)
                plan.launch(); ctx.synchronize()
                plan.timing(True)
                for _ in range(args.launches):
                    plan.launch()
                ms, k = plan.elapsed()
                rec = plan.fetch().tobytes()
                if label == "P=all":
                    ref = ref or rec
                    assert rec == ref, "records depend on the tile size"
                row = {"n_hap": n, "windows": n_win, "window_sites": W, "tile_blocks": tb, "tiles": plan.n_tiles, "masks": label,
                       "kernel_ms": ms / k, "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9,
                       "frac_of_8TBps": plan.bytes_streamed / (ms / k / 1e3) / 8e12}
                rows.append(row)
                print(json.dumps(row), flush=True)
                plan.destroy()
        bm.free()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": "any-n scan kernel, 1x MI355X, HIP-event kernel time", "rows": rows}, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
