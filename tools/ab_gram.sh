# A/B timing of gram_fp4_kernel builds / launch shapes on 4096 x (465 hap x 50 kb) windows; run on the GPU box from the repo root.
# usage: bash tools/ab_gram.sh "tag:ENV=VAL,ENV2=VAL2[:variant]" ...   (variant = impop_amd/_variants/libimpop_<variant>.so)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; var=""; [ "$rest" != "$envs" ] && var=${rest#*:}
  ( for kv in ${envs//,/ }; do [ -n "$kv" ] && export "$kv"; done
    [ -n "$var" ] && export IMPOP_HIP_LIBRARY=$R/impop_amd/_variants/libimpop_$var.so
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_$tag -- python3 $R/tools/bench_pairwise.py --windows ${AB_WINDOWS:-4096} --no-check --big-sites 200000 > $R/gpurun_out/ab_$tag.json 2> $R/gpurun_out/ab_$tag.err || tail -5 $R/gpurun_out/ab_$tag.err )
  python3 - $R/gpurun_out/ab_$tag $tag <<'PY'
import csv, glob, json, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "gram_fp4" in r["Kernel_Name"]]
j = json.load(open(sys.argv[1] + ".json"))
print(sys.argv[2], "gram ms:", " ".join(f"{x:.2f}" for x in d[:5]), "| scan w/s %.0f" % j["pairwise_scan_465x50kb"]["windows_per_s"],
      "| compacted w/s %.0f" % j["pairwise_scan_465x50kb_variable_sites_only"]["windows_per_s"], flush=True)
for k in ("hfst_kernel", "pica2_kernel"):
    e = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if k in r["Kernel_Name"]]
    print("   ", k, "us:", " ".join(f"{x:.0f}" for x in e), flush=True)
PY
done
