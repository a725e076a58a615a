#!/bin/bash
# the default bench (no side figures, no CPU baseline) with every library variant under tools/variants: registration
# milliseconds and the staging kernel's
for f in tools/variants/*.so; do
  MISTITCH_LIB=$PWD/$f timeout -k 10 300 python bench.py --cpu-pairs 0 --no-side 2>/dev/null | tail -n 1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$f', j['value'], j['registration']['ms_per_step'], j['kernels']['k_stage_strips']['ms_per_step'])" || exit 1
done
