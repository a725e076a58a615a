#!/bin/bash
# Run on the GPU box: counter passes over the default bench workload for one kernel's analysis.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-side --no-profile"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq1 -o x -- $B > $R/gpurun_out/pmc_sq1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 --output-format csv -d $R/gpurun_out/pmc_sq2 -o x -- $B > $R/gpurun_out/pmc_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $R/gpurun_out/pmc_mem -o x -- $B > $R/gpurun_out/pmc_mem.log 2>&1
cd $R
python3 tools/pmc_kernel_counters.py --match "$1" gpurun_out/pmc_sq1/x_counter_collection.csv gpurun_out/pmc_sq2/x_counter_collection.csv gpurun_out/pmc_mem/x_counter_collection.csv > gpurun_out/pmc_$1.txt
cat gpurun_out/pmc_$1.txt
