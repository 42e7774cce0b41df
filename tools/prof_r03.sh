# Round-3 evidence in one call (run on the GPU box from the repo root): bash tools/prof_r03.sh r03a
#   full GPU suite, smoke, the default bench command under rocprofv3 --kernel-trace --stats, the two PMC traffic passes of the headline
#   kernel, bench --secondary, and the all-pairs path: kernel stats, MFMA PMC pass, FETCH_SIZE pass (polarity on and off)
export TMPDIR=/tmp
TAG=${1:-r03}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o p -- python3 bench.py > $O/bench_n1.json 2> $O/bench_stats.err
cut -c1-400 $O/bench_n1.json; echo
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_fetch.out 2> $O/bench_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_write.out 2> $O/bench_write.err
python3 tools/summarise_pmc_traffic.py $O/bench_fetch/p_counter_collection.csv $O/bench_write/p_counter_collection.csv $O/pmc_hbm_traffic.json
timeout -k 10 600 python3 bench.py --secondary --no-cpu-baseline > $O/bench_secondary.json 2> $O/bench_secondary.err
cut -c1-200 $O/bench_secondary.json; echo
# ---- all-pairs path
timeout -k 10 300 python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pairwise_bench.json 2> $O/pairwise_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_stats -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_stats.out 2> $O/pw_stats.err
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pw_mfma -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_mfma.out 2> $O/pw_mfma.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pw_fetch -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_fetch.out 2> $O/pw_fetch.err
python3 tools/summarise_pmc_gram.py $O/pw_mfma/p_counter_collection.csv $O/pw_fetch/p_counter_collection.csv $O/pairwise_pmc.json 4096 2906250
IMPOP_NO_POLARITY=1 timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pw_mfma_nopol -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_mfma_nopol.out 2> $O/pw_mfma_nopol.err
IMPOP_NO_POLARITY=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pw_fetch_nopol -o p -- python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pw_fetch_nopol.out 2> $O/pw_fetch_nopol.err
python3 tools/summarise_pmc_gram.py $O/pw_mfma_nopol/p_counter_collection.csv $O/pw_fetch_nopol/p_counter_collection.csv $O/pairwise_pmc_nopol.json 4096 2906250
IMPOP_NO_POLARITY=1 timeout -k 10 300 python3 tools/bench_pairwise.py --windows 4096 --big-sites 200000 > $O/pairwise_bench_nopol.json 2> $O/pairwise_bench_nopol.err
ls $O
