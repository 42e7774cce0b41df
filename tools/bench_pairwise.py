#!/usr/bin/env python3
"""All-pairs path measurement (not the bench line): Gram kernel + pica2/h-fst epilogues on
BASELINE config shapes.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times.

    python tools/bench_pairwise.py [--windows 64] [--big-sites 1000000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VALU_PEAK_LANEOPS = 256 * 128 * 2.4e9  # 256 CUs x 128 lane-ops/clk x 2.4 GHz (MI355X_MICROARCH.md)
FP4_DENSE_PEAK_MACS = 5.0e15           # ~10 PFLOP/s dense FP4 (MI355X_MICROARCH.md) = 5e15 MAC/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=256)
    ap.add_argument("--window", type=int, default=50000)
    ap.add_argument("--big-hap", type=int, default=4096)
    ap.add_argument("--big-sites", type=int, default=1_000_000)
    ap.add_argument("--no-check", action="store_true", help="ablation builds produce wrong results")
    args = ap.parse_args()
    import numpy as np
    import impop_amd
    from oracle import oracle as orc

    ctx = impop_amd.Context(0)
    out = {}
    # ---- configs 2/3 shape: 465 haplotypes, 50 kb windows, thresholded pica2 (-t 0.999 -r 5) + h-fst
    n, W, NW = 465, args.window, args.windows
    bm = ctx.synthetic(n, W * NW, seed=20251031, keep_hap_major=True)
    wins = impop_amd.fixed_windows(W * NW, W)
    in_a = np.zeros(n, np.uint8); in_a[:140] = 1
    in_b = np.zeros(n, np.uint8); in_b[140:240] = 1
    bm.pairwise_scan(wins, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5, s_scope=2)  # scratch + code objects, no S
    ctx.synchronize()
    t0 = time.perf_counter()  # first call that needs S (the site bitmap itself was built with the matrix)
    res = bm.pairwise_scan(wins, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5)
    dt_first = time.perf_counter() - t0
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = bm.pairwise_scan(wins, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5)
    dt = (time.perf_counter() - t0) / reps
    # parity of one window against the oracle (dense functions)
    bits = bm.download(0, W)
    sim = orc.identity(orc.pairwise_counts(bits, n, 0, W), W, 0)
    pi, ps, _, G = orc.pica2(sim, 0.999, W, 5)
    assert args.no_check or (abs(float(res[0]["pi"]) - pi) <= 1e-9 * abs(pi) and int(res[0]["n_groups"]) == G), (res[0], pi, G)
    pair_words = (n * (n + 1) // 2) * ((W + 31) // 32)
    macs = n * (n + 1) // 2 * W  # algorithmic MACs per window (upper triangle incl. diagonal)
    out["pairwise_scan_465x50kb"] = {"windows": NW, "s_per_batch": dt, "windows_per_s": NW / dt, "groups_window0": G,
                                     "first_call_with_S_s": dt_first, "first_call_windows_per_s": NW / dt_first,
                                     "algorithmic_macs_per_window": macs, "algorithmic_macs_per_s": macs * NW / dt,
                                     "frac_of_fp4_dense_peak_end_to_end": macs * NW / dt / FP4_DENSE_PEAK_MACS,
                                     "gram_kernel": os.environ.get("IMPOP_GRAM_MFMA", "fp4")}
    # the same windows from the matrix compacted to its variable sites (impop_matrix_compact): identical records,
    # the contraction runs over the kept sites only (+ the per-window count of dropped all-ones sites)
    t0 = time.perf_counter()
    cm = bm.compact()
    ctx.synchronize()
    t_compact = time.perf_counter() - t0
    rc = cm.pairwise_scan(wins, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5)
    same = rc.tobytes() == res.tobytes()
    t0 = time.perf_counter()
    for _ in range(reps):
        cm.pairwise_scan(wins, None, in_a, in_b, kind="match", threshold=0.999, round_digits=5)
    dtc = (time.perf_counter() - t0) / reps
    out["pairwise_scan_465x50kb_variable_sites_only"] = {"windows": NW, "s_per_batch": dtc, "windows_per_s": NW / dtc, "kept_sites": cm.n_site,
                                                         "of_sites": bm.n_site, "compaction_s": t_compact,
                                                         "records_identical_to_full_matrix": bool(same)}
    cm.free()
    bm.free()
    # ---- config 5 shape: 4096 haplotypes, one long window, integer Gram only
    nb, Wb = args.big_hap, args.big_sites
    bm = ctx.synthetic(nb, Wb, seed=5, n_founder=16, p_founder=0.05, p_private_word=0.05, keep_hap_major=True)
    I = bm.pairwise_counts(0, Wb)
    ctx.synchronize()
    t0 = time.perf_counter()
    I = bm.pairwise_counts(0, Wb)
    dt = time.perf_counter() - t0
    # spot-check rows against numpy on the downloaded bits
    m = impop_amd.unpack_hap_major(bm.download(0, min(Wb, 200000)), min(Wb, 200000)).astype(np.int64)
    Ic = bm.pairwise_counts(0, min(Wb, 200000))
    idx = [0, 1, 63, 64, 127, 128, 1000, nb - 1]
    want = m[idx] @ m.T
    assert args.no_check or (Ic[idx].astype(np.int64) == want).all()
    pair_words = (nb * (nb + 1) // 2) * ((Wb + 31) // 32)
    macs_b = nb * (nb + 1) // 2 * Wb
    out["gram_%dx%d" % (nb, Wb)] = {"s_incl_copy_out": dt, "algorithmic_macs": macs_b, "macs_per_s_incl_copy": macs_b / dt,
                                    "frac_of_fp4_dense_peak_incl_copy": macs_b / dt / FP4_DENSE_PEAK_MACS}
    print(json.dumps(out))
    bm.free()
    ctx.close()


if __name__ == "__main__":
    main()
