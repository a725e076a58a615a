#!/bin/bash
# Run on the GPU box: bytes that k_fuse fetches and writes beyond L2 per launch (FETCH_SIZE / WRITE_SIZE passes over
# tools/fuse_probe.py), for every blend mode with and without a flatfield; the template arguments in the kernel
# names tell the variants apart.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/fuse_probe.py --reps 2"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $R/gpurun_out/ft_fetch -o x -- $P > $R/gpurun_out/ft_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/ft_write -o x -- $P > $R/gpurun_out/ft_write.log 2>&1
cd $R
python3 tools/pmc_kernel_counters.py --match k_fuse $(find gpurun_out/ft_fetch gpurun_out/ft_write -name '*counter_collection.csv')
