#!/bin/bash
# Counter passes over the fusion probe (one variant), one rocprofv3 run per pass.
# usage: tools/pmc_fuse_rows.sh <variant, e.g. overwrite:f32>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
V=${1:-overwrite:f32}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/fuse_probe.py --reps 2 --only $V"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_a -o x -- $B > $R/gpurun_out/pmc_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/pmc_b -o x -- $B > $R/gpurun_out/pmc_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc_c -o x -- $B > $R/gpurun_out/pmc_c.log 2>&1
python3 - $R <<'PY'
import csv, sys, glob, collections
R = sys.argv[1]
for d in ("pmc_a", "pmc_b", "pmc_c"):
    for path in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "k_fuse" not in k: continue
            agg[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in agg.items():
            print(d, k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in dd.items()))
PY
