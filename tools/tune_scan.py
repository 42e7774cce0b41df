#!/usr/bin/env python3
"""Tuning harness for the streaming scan kernel (not part of the product path).

    python tools/tune_scan.py build      # here (CPU container): hipcc every variant
    python tools/tune_scan.py run        # on the GPU box: interleaved A/B in ONE process

Variants = compile-time knobs of impop_amd/csrc/scan.hip (IMPOP_SCAN_NT, IMPOP_SCAN_UNROLL,
IMPOP_SCAN_MIN_WAVES) x the runtime tile size.  Each variant is its own .so, loaded side by
side with ctypes (RTLD_LOCAL), timed with the library's own HIP events, rounds interleaved.
"""
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "impop_amd", "_variants")
VARIANTS = [(nt, u, w) for nt in (0, 1) for u in (1, 2, 4) for w in (1,)] + [(0, 2, 6), (1, 2, 6), (0, 2, 8), (1, 1, 8)]


def so_name(v):
    return os.path.join(VDIR, "libimpop_hip_nt%d_u%d_w%d.so" % v)


def build():
    from impop_amd import build as b
    os.makedirs(VDIR, exist_ok=True)

    def one(v):
        cmd = ["hipcc"] + b.FLAGS + ["-DIMPOP_SCAN_NT=%d" % v[0], "-DIMPOP_SCAN_UNROLL=%d" % v[1],
                                     "-DIMPOP_SCAN_MIN_WAVES=%d" % v[2], "-o", so_name(v)] + \
              [os.path.join(b.CSRC, s) for s in b.SOURCES]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return v, r.returncode, r.stderr[-400:]
    with ThreadPoolExecutor(4) as ex:
        for v, rc, err in ex.map(one, VARIANTS):
            print(v, "ok" if rc == 0 else "FAILED " + err, flush=True)


def run(n_windows=1500, window=50000, n_hap=465, rounds=5, launches=4):
    from impop_amd import _lib
    from impop_amd._lib import ScanParams, SynthParams, Window
    import numpy as np
    import impop_amd
    _lib._preload_hip_runtime()
    wins = impop_amd.fixed_windows(n_windows * window, window)
    in_a = np.zeros(n_hap, np.uint8); in_a[:140] = 1
    in_b = np.zeros(n_hap, np.uint8); in_b[140:240] = 1
    ma, mb = impop_amd.pack_mask(in_a, n_hap), impop_amd.pack_mask(in_b, n_hap)
    u64p = C.POINTER(C.c_uint64)
    configs = []
    ref = None
    for v in VARIANTS:
        if not os.path.exists(so_name(v)):
            continue
        lib = C.CDLL(so_name(v))
        for name, (res, args) in _lib.SIGNATURES.items():
            f = getattr(lib, name); f.restype = res; f.argtypes = args
        ctx = C.c_void_p()
        assert lib.impop_ctx_create(0, None, C.byref(ctx)) == 0, lib.impop_last_error()
        sp = SynthParams(20251031, 8, 1e-3, 3.2e-3)
        m = C.c_void_p()
        assert lib.impop_matrix_synthetic(ctx, n_hap, n_windows * window, C.byref(sp), 1, C.byref(m)) == 0, lib.impop_last_error()
        for tb in (16, 32, 64, 128, 256):
            prm = ScanParams(C.sizeof(ScanParams), 0, 0, tb)
            plan = C.c_void_p()
            assert lib.impop_scan_plan_create(ctx, m, wins.ctypes.data_as(C.POINTER(Window)), len(wins), None,
                                              ma.ctypes.data_as(u64p), mb.ctypes.data_as(u64p), C.byref(prm), C.byref(plan)) == 0
            # correctness of every variant against the first one (integers + doubles bit-identical)
            lib.impop_scan_plan_launch(plan, None)
            out = np.zeros(len(wins), dtype=impop_amd.STATS_DTYPE)
            lib.impop_scan_plan_fetch(plan, out.ctypes.data_as(C.POINTER(_lib.WindowStats)))
            if ref is None:
                ref = out.tobytes()
            assert out.tobytes() == ref, ("variant result differs", v, tb)
            configs.append({"variant": v, "tile_blocks": tb, "lib": lib, "plan": plan, "ms": []})
    algo = n_hap * n_windows * window / 8.0
    for r in range(rounds):
        for c in configs:
            lib, plan = c["lib"], c["plan"]
            lib.impop_scan_plan_timing(plan, 1)
            for _ in range(launches):
                lib.impop_scan_plan_launch(plan, None)
            t, k = C.c_double(), C.c_uint64()
            lib.impop_scan_plan_elapsed(plan, C.byref(t), C.byref(k))
            lib.impop_scan_plan_timing(plan, 0)
            c["ms"].append(t.value / k.value)
    res = []
    for c in configs:
        med, best = statistics.median(c["ms"]), min(c["ms"])
        res.append({"nt": c["variant"][0], "unroll": c["variant"][1], "min_waves": c["variant"][2], "tile_blocks": c["tile_blocks"],
                    "ms_median": med, "ms_min": best, "algo_GBps_median": algo / med / 1e6, "frac_of_8TBps": algo / med / 1e6 / 8000})
    res.sort(key=lambda r: r["ms_median"])
    for r in res:
        print(json.dumps(r))
    return res


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        run()
