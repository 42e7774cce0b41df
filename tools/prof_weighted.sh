# per-kernel durations of tools/bench_weighted.py (node-level matrix, node lengths as weights); run on the GPU box from
# the repo root
set -e
export TMPDIR=/tmp
O=gpurun_out/weighted
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 tools/bench_weighted.py > $O/bench.json 2> $O/prof.err
cat $O/bench.json
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/prof/**/p_kernel_trace.csv", recursive=True))[-1]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("impop::", "").replace("void ", "")
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.5:
        print(f"{k:40s} n={len(v):4d} total={sum(v):8.2f} ms  mean={sum(v)/len(v):.3f}  first:", " ".join(f"{x:.2f}" for x in v[:14]))
PY
