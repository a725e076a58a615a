#!/bin/bash
# Run HERE after `gpurun -- bash tools/final_gpu_pass.sh`: summarise the counter passes and copy the round's figures
# from gpurun_out/ into profiles/ (tracked).  usage: refresh_profiles.sh <tag, e.g. r3>
set -e
tag=${1:-r3}
python tools/pmc_summary.py --tag $tag --workload c3 --fft f64 --stats gpurun_out/prof/x_kernel_stats.csv \
  --pmc gpurun_out/pmc_fetch/x_counter_collection.csv gpurun_out/pmc_write/x_counter_collection.csv
tail -n 1 gpurun_out/bench_c3_final.log > profiles/${tag}_bench_c3.json
tail -n 1 gpurun_out/bench_c2.log > profiles/${tag}_bench_c2.json
tail -n 1 gpurun_out/bench_e195.log > profiles/${tag}_bench_c3_e195.json
cp gpurun_out/rank_of_8.json profiles/${tag}_rank_of_8.json
cp gpurun_out/gputest.txt profiles/${tag}_gputest.txt
