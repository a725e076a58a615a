#!/bin/bash
# Run on the GPU box: one SQ counter pass (wave cycles, waits, VALU / LDS activity, LDS conflicts) over the
# default bench workload, summarised per kernel for the names in $1 (comma separated substrings).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-side --no-profile"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -o x -- $B > $R/gpurun_out/pmc_sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq2 -o x -- $B > $R/gpurun_out/pmc_sq2.log 2>&1
cd $R
for k in ${1//,/ }; do
  python3 tools/pmc_kernel_counters.py --match $k $(find gpurun_out/pmc_sq gpurun_out/pmc_sq2 -name '*counter_collection.csv')
done
