#!/bin/bash
# Run on the GPU box: the GPU test suite, the kernel statistics and counter passes of the default bench, the side
# bench lines and the rank-of-8 projection, every step under its own time limit; results under gpurun_out/.
# The bench line kept for profiles/ is a second run AFTER tools/pmc_summary.py would have stamped the counters, so this
# script stamps them itself first (same command the builder runs here afterwards).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputest.txt 2>&1
bash tools/profile_bench.sh
python3 tools/pmc_summary.py --tag r3 --workload c3 --fft f64 --stats gpurun_out/prof/x_kernel_stats.csv \
  --pmc gpurun_out/pmc_fetch/x_counter_collection.csv gpurun_out/pmc_write/x_counter_collection.csv > gpurun_out/pmc_summary.log 2>&1
timeout -k 10 600 python3 bench.py > gpurun_out/bench_c3_final.log 2>&1
timeout -k 10 300 python3 bench.py --workload c2 --cpu-pairs 0 > gpurun_out/bench_c2.log 2>&1
timeout -k 10 300 python3 bench.py --edge-width 195 --cpu-pairs 0 > gpurun_out/bench_e195.log 2>&1
bash tools/rank_of_8.sh
tail -n 2 gpurun_out/gputest.txt
tail -n 1 gpurun_out/bench_c3_final.log | cut -c1-300
