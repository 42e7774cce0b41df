set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_stats -o p -- python3 tools/bench_pairwise.py --windows 4096 > $O/pairwise_bench.json 2> $O/pw_stats.err
echo "stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pw_mfma -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_mfma.out 2> $O/pw_mfma.err
echo "mfma pmc done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pw_fetch -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_fetch.out 2> $O/pw_fetch.err
echo "fetch pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o p -- python3 tools/bench_config5.py > $O/config5.log 2> $O/c5_stats.err
echo "config5 stats done"
find $O -name "*.csv" | head -30
