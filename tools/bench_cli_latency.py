#!/usr/bin/env python3
"""Per-window latency of the drop-in CLIs on one 465-haplotype `.sim` file (the unit the reference's
bash drivers fork per window): wall time of `python scripts/pica2.py` / `h-fst.py` as subprocesses,
next to the reference-style pure-Python chain (oracle/ref_style.py) on the same file in-process.
Prints one JSON object; not the bench line."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import impop_amd
from oracle import ref_style

n, W = 465, 50000
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, W, seed=20251031, keep_hap_major=True)
sim = bm.pairwise_identity(0, W, "match")
bm.free(); ctx.close()
names = [f"H{i // 2:05d}#{i % 2 + 1}#chr2:0-{W}" for i in range(n)]
out = {"n_hap": n, "rows": n * n}
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "win.sim")
    text = ref_style.write_sim(names, sim)
    open(p, "w").write(text)
    out["sim_bytes"] = len(text)
    pa, pb = os.path.join(td, "A.txt"), os.path.join(td, "B.txt")
    open(pa, "w").write("\n".join(f"H{i:05d}" for i in range(0, 70)) + "\n")
    open(pb, "w").write("\n".join(f"H{i:05d}" for i in range(70, 120)) + "\n")

    def wall(argv):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable] + argv, capture_output=True, text=True, cwd=td)
            ts.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr
        return {"first_s": ts[0], "best_of_next_3_s": min(ts[1:]), "stdout": r.stdout.strip()}

    sc = os.path.join(ROOT, "scripts")
    out["pica2_cli"] = wall([os.path.join(sc, "pica2.py"), p, "-t", "0.999", "-r", "5", "-l", str(W), "-d", td])
    out["hfst_cli"] = wall([os.path.join(sc, "h-fst.py"), p, "-a", pa, "-b", pb, "-l", str(W), "-d", td])
    t0 = time.perf_counter()
    table, elements, rows = ref_style.parse_sim(open(p).read())
    t_parse = time.perf_counter() - t0
    t0 = time.perf_counter()
    pi, ps = ref_style.pica2_pi(table, elements, 0.999, W, 5)
    t_pi = time.perf_counter() - t0
    out["reference_style_python"] = {"parse_s": t_parse, "pica2_s": t_pi, "total_s": t_parse + t_pi,
                                     "pi_site": f"{ps:.8f}"}
print(json.dumps(out))
