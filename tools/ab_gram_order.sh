R=$(pwd)
bash tools/ab_gram.sh "prod::" "ord1::ord1" "ord2::ord2" 2>&1 | grep "gram ms"
export TMPDIR=/tmp
for v in prod ord1 ord2; do
  ( [ "$v" != prod ] && export IMPOP_HIP_LIBRARY=$R/impop_amd/_variants/libimpop_$v.so
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/abf_$v -o p -- python3 $R/tools/bench_pairwise.py --windows 4096 --no-check --big-sites 200000 > /dev/null 2> $R/gpurun_out/abf_$v.err )
  python3 - $R/gpurun_out/abf_$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if "gram_fp4" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if us > 4000: out.append((round(us), round(float(r["Counter_Value"]) * 2048 / 4096 / 2906250, 2)))
print(sys.argv[2], "gram (us, x window bytes):", out)
PY
done
