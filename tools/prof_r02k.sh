# rocprofv3 evidence for the end-of-round state (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or pairwise or hud or identity or compacted or weighted or pica2" > $O/pytest_subset.log 2>&1 || { tail -20 $O/pytest_subset.log; exit 1; }
tail -2 $O/pytest_subset.log
python3 tools/bench_config5.py --out $O/config5.json > $O/config5.log 2>&1
grep what $O/config5.log | cut -c1-120
# the exact default bench command under the profiler (kernel stats), then the two PMC traffic passes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o p -- python3 bench.py > $O/bench_n1.json 2> $O/bench_stats.err
cat $O/bench_n1.json | cut -c1-400
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fetch.out 2> $O/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_write.out 2> $O/bench_write.err
python3 tools/summarise_pmc_traffic.py $O/bench_fetch/p_counter_collection.csv $O/bench_write/p_counter_collection.csv $O/pmc_hbm_traffic.json
