# GPU timeline (kernels + copies) of one impop_pairwise_scan call on the bench shape; run on the GPU box from the repo root
export TMPDIR=/tmp
O=gpurun_out/timeline
mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o p -- python3 tools/trace_pairwise_host.py > $O/run.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
ev = []
for f in glob.glob(o + "/prof/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("impop::", "")))
for f in glob.glob(o + "/prof/**/p_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the last two calls: everything after the third-last gram_fp4_kernel
grams = [i for i, e in enumerate(ev) if e[2].startswith("gram_fp4")]
for gi in grams[-4:]:
    lo = gi
    while lo > 0 and ev[gi][0] - ev[lo - 1][1] < 400_000: lo -= 1
    hi = gi
    while hi + 1 < len(ev) and ev[hi + 1][0] - ev[gi][1] < 1_500_000 and not ev[hi + 1][2].startswith("gram_fp4"): hi += 1
    t0 = ev[lo][0]
    print("---- call around gram dispatch", gi)
    for s, e, n in ev[lo:hi + 1]:
        print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:9.1f} us  {n}")
PY
