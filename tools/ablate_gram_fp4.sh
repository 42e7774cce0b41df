# Timing-only ablation of gram_fp4_kernel (results of variants 1-3 are wrong by construction): which of expansion VALU /
# global loads costs the kernel its clock?  Run on the GPU box from the repo root: bash tools/ablate_gram_fp4.sh
# variants (python tools/build_variants.py pairwise.hip ab1:-DIMPOP_GRAM_ABLATE=1 ab2:-DIMPOP_GRAM_ABLATE=2 ab3:-DIMPOP_GRAM_ABLATE=3):
#   0 product, 1 no in-loop expansion VALU, 2 no in-loop global loads, 3 neither (MFMAs + task framing only),
#   4 loads + VALU both run but the expansions read a cell that is never reloaded (no wait on load data)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 3 4}; do
  if [ $v = 0 ]; then unset IMPOP_HIP_LIBRARY; else export IMPOP_HIP_LIBRARY=$R/impop_amd/_variants/libimpop_ab$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abf$v -- python3 $R/tools/bench_pairwise.py --windows 4096 --no-check --big-sites 200000 > $R/gpurun_out/abf$v.json 2> $R/gpurun_out/abf$v.err || { tail -5 $R/gpurun_out/abf$v.err; exit 1; }
  python3 - $R/gpurun_out/abf$v $v <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "gram_fp4" in r["Kernel_Name"]]
print("variant", sys.argv[2], "gram_fp4_kernel ms per launch:", " ".join(f"{x:.2f}" for x in d))
PY
done
