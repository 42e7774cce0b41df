# rocprofv3 evidence for the final tree of a round (run on the GPU box from the repo root): bash tools/prof_final.sh r02zd
set -e
export TMPDIR=/tmp
TAG=${1:-final}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
# the exact default bench command under the profiler (kernel stats), then the two PMC traffic passes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o p -- python3 bench.py > $O/bench_n1.json 2> $O/bench_stats.err
cut -c1-600 $O/bench_n1.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fetch.out 2> $O/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_write.out 2> $O/bench_write.err
python3 tools/summarise_pmc_traffic.py $O/bench_fetch/p_counter_collection.csv $O/bench_write/p_counter_collection.csv $O/pmc_hbm_traffic.json
python3 bench.py --secondary --no-cpu-baseline > $O/bench_secondary.json 2> $O/bench_secondary.err
cut -c1-300 $O/bench_secondary.json
