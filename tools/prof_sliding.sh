# per-kernel durations of tools/time_sliding.py (4096 x 465-haplotype 10 kb windows at a 5 kb step); run on the GPU box from the repo root
export TMPDIR=/tmp
O=gpurun_out/slide
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 tools/time_sliding.py > $O/time.log 2> $O/prof.err
grep sliding $O/time.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/prof/**/p_kernel_trace.csv", recursive=True))[-1]
per = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("impop::", "").replace("void ", "")
    per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.3:
        print(f"{k:36s} n={len(v):3d} per launch:", " ".join(f"{x:.2f}" for x in v[:12]))
PY
