"""All-core run of the reference-STYLE chain (oracle/ref_style.py) for bench.py's cpu_baseline: the way the
reference scales today — one Python process per window, `multiprocessing` over the host's cores.  Runs in
its own interpreter (bench.py starts it as a child process: the bench process itself holds the GPU).
TEST / BENCH INFRASTRUCTURE ONLY, like everything under oracle/.

    python -m oracle.ref_style_mp windows.npz n_procs   ->  one JSON line
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_G = {}


def _one(k):
    from oracle import ref_style
    names, sims, in_a, in_b, L, S = _G["names"], _G["sims"], _G["in_a"], _G["in_b"], _G["L"], _G["S"]
    r = ref_style.window_chain(names, sims[k % len(sims)], in_a, in_b, L, S[k % len(sims)])
    return r["pi_site"]


def main():
    from oracle import oracle as orc
    z = np.load(sys.argv[1], allow_pickle=False)
    n_procs = int(sys.argv[2])
    n, W = int(z["n"]), int(z["W"])
    in_a, in_b = z["in_a"], z["in_b"]
    ones = orc.pack_mask(np.ones(n, np.uint8))
    sims, S = [], []
    for bits in z["bits"]:  # identity tables (what `impg similarity` would hand over): not timed
        sims.append(orc.identity(orc.pairwise_counts(bits, n, 0, W), W, 0))
        S.append(orc.window_sitecount(bits, n, 0, W, ones, orc.pack_mask(in_a), orc.pack_mask(in_b), W)["s_all"])
    _G.update(names=[f"H{i // 2:05d}#{i % 2 + 1}#chr2:0-{W}" for i in range(n)], sims=sims, in_a=in_a, in_b=in_b, L=W, S=S)
    jobs = max(2 * n_procs, len(sims))
    with mp.get_context("fork").Pool(n_procs) as pool:  # fork: the tables are inherited, nothing is pickled per job
        pool.map(_one, range(n_procs))  # warm-up (imports)
        t0 = time.perf_counter()
        out = pool.map(_one, range(jobs), chunksize=1)
        dt = time.perf_counter() - t0
    print(json.dumps({"value": jobs / dt, "unit": "windows/s", "cores": n_procs,
                      "sample": f"{jobs} windows ({len(sims)} distinct) through oracle/ref_style.py, multiprocessing over {n_procs} processes, {dt:.1f} s",
                      "pi_site_first": out[0]}))


if __name__ == "__main__":
    main()
