#!/bin/bash
# registration step time against pairs per pass (mistitch_set_pair_chunk) on the default workload
for c in ${@:-8 16 32 64 128 256}; do
  timeout -k 10 150 python bench.py --no-side --cpu-pairs 0 --no-profile --pair-chunk $c > gpurun_out/chunk_$c.log 2>&1 || exit 1
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/chunk_$c.log") if l.startswith('{"metric"')][0]
print($c, d["value"], d["registration"]["ms_per_step"])
PY
done
