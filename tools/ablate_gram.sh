cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4 7; do
  if [ $v = 0 ]; then unset IMPOP_HIP_LIBRARY; else export IMPOP_HIP_LIBRARY=$R/impop_amd/_variants/libimpop_ab$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab$v -- python3 $R/tools/bench_pairwise.py --windows 512 --no-check --big-sites 200000 > $R/gpurun_out/ab$v.json 2> $R/gpurun_out/ab$v.err || exit 1
  echo "variant $v: $(grep gram_mfma $R/gpurun_out/ab$v/*/*_kernel_stats.csv | cut -d, -f2-7)"
done
