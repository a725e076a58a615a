#!/bin/bash
# Run on the GPU box: kernel statistics and the two HBM traffic passes of the default bench
# workload, each under its own time limit; output under gpurun_out/ for tools/pmc_summary.py.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-side"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o x -- $B > $R/gpurun_out/prof.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o x -- $B > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o x -- $B > $R/gpurun_out/pmc_write.log 2>&1
cd $R && timeout -k 10 600 python3 bench.py > gpurun_out/bench_c3.log 2>&1
tail -n 1 gpurun_out/bench_c3.log | cut -c1-400
