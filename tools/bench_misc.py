#!/usr/bin/env python3
"""Secondary measurements (not the bench line): host->HBM upload rate incl. the layout transposes,
K-population Fst pass, AFS pass, sliding-window scan (BASELINE config 4 shape)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import impop_amd
    ctx = impop_amd.Context(0)
    out = {}
    n = 465
    # upload: 465 x 20 M sites = 1.16 GB of host bits
    W = 20_000_000
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 63, size=(n, W // 64), dtype=np.uint64)
    t0 = time.perf_counter()
    bm = ctx.upload(bits, W, keep_hap_major=False)
    dt = time.perf_counter() - t0
    out["upload"] = {"host_bytes": bits.nbytes, "seconds": dt, "GBps": bits.nbytes / dt / 1e9,
                     "windows_50kb_per_s_pcie_inclusive": (W / 50000) / dt}
    bm.free()
    del bits
    # config 4 shape: 10 kb windows, 5 kb step (every site belongs to two windows but is read once)
    NW = 60000
    W = 5000 * (NW + 1)
    bm = ctx.synthetic(n, W, seed=4)
    wins = impop_amd.fixed_windows(W, 10000, 5000)
    in_a = np.zeros(n, np.uint8); in_a[:140] = 1
    in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
    plan = bm.plan(wins, None, in_a, in_b)
    plan.launch(); ctx.synchronize()
    plan.timing(True)
    for _ in range(10):
        plan.launch()
    ms, k = plan.elapsed()
    out["sliding_10kb_step_5kb"] = {"windows": len(wins), "kernel_ms": ms / k, "windows_per_s": len(wins) / (ms / k / 1e3),
                                    "bytes_streamed": plan.bytes_streamed, "layout_GBps": plan.bytes_streamed / (ms / k / 1e3) / 1e9}
    plan.destroy()
    # K = 5 populations, all 10 pairs, 50 kb windows
    w50 = impop_amd.fixed_windows(W, 50000)
    sizes = [140, 88, 100, 60, 72]
    pops, o = [], 0
    for s in sizes:
        f = np.zeros(n, np.uint8); f[o:o + s] = 1; o += s
        pops.append(f)
    bm.scan_multi(w50[:10], pops)
    ctx.synchronize()
    t0 = time.perf_counter()
    r = bm.scan_multi(w50, pops)
    dt = time.perf_counter() - t0
    out["scan_multi_K5"] = {"windows": len(w50), "pairs": 10, "seconds_incl_plan_and_copy": dt, "windows_per_s": len(w50) / dt,
                            "matrix_GB": bm.device_bytes / 1e9, "layout_GBps_incl_overheads": bm.device_bytes / dt / 1e9}
    # AFS
    bm.afs(w50[:10])
    t0 = time.perf_counter()
    a = bm.afs(w50)
    dt = time.perf_counter() - t0
    assert int(a.sum()) == int(sum(int(x["site_end"]) - int(x["site_begin"]) for x in w50))
    out["afs"] = {"windows": len(w50), "seconds_incl_copy": dt, "windows_per_s": len(w50) / dt}
    print(json.dumps(out))
    bm.free(); ctx.close()


if __name__ == "__main__":
    main()
