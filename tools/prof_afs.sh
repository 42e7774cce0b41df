# kernel time of impop_afs on the chr2-scale workload (465 haplotypes, 4854 x 50 kb windows); run on the GPU box from the repo root
set -e
export TMPDIR=/tmp
O=gpurun_out/afs
mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, impop_amd
n, W, NW = 465, 50000, 4854
ctx = impop_amd.Context(0)
bm = ctx.synthetic(n, W * NW, seed=20251031)
wins = impop_amd.fixed_windows(W * NW, W)
bm.afs(wins[:10])
for _ in range(3):
    t0 = time.perf_counter(); a = bm.afs(wins); dt = time.perf_counter() - t0
    print("afs wall s", round(dt, 4), "windows/s incl copy", round(NW / dt))
assert int(a.sum()) == W * NW
t0 = time.perf_counter(); a1 = bm.afs(wins[:1]); print("one window wall ms", round((time.perf_counter() - t0) * 1e3, 3))
big = impop_amd.fixed_windows(W * NW, W * NW)
t0 = time.perf_counter(); ab = bm.afs(big); print("one chr-long window wall ms", round((time.perf_counter() - t0) * 1e3, 3))
assert (ab[0] == a.sum(axis=0)).all()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $O/run.py > $O/out.log 2> $O/err.log
grep -v amdgpu $O/out.log
grep afs_kernel $O/prof/p_kernel_stats.csv | cut -d, -f1-8 | cut -c1-40,160-
python3 - "$O" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + "/prof/p_kernel_trace.csv")) if "afs_kernel" in r["Kernel_Name"]]
print("afs_kernel ms per launch:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3) for r in rows])
PY
