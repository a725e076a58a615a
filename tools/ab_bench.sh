#!/bin/bash
# Same-box comparison of builds of the library: tools/ab_bench.sh <a.so> <b.so> ... [-- bench args]
# Takes them in turn, twice, and prints the per-kernel milliseconds of each run.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for round in 1 2; do
  for lib in "${LIBS[@]}"; do
    MISTITCH_LIB=$R/$lib timeout -k 10 200 python3 $R/bench.py --steps 5 --cpu-pairs 0 --no-side "$@" > /tmp/ab.json 2> /tmp/ab.err
    python3 - "$lib" <<'PY'
import json, sys
l = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = l['kernels']
print(sys.argv[1].split('/')[-1], 'reg %.3f fuse %.3f |' % (l['registration']['ms_per_step'], l['fusion']['ms_per_step']),
      ' '.join('%s %.3f' % (n.replace('k_', ''), v['ms_per_step']) for n, v in k.items() if v['ms_per_step'] > 0.2), flush=True)
PY
  done
done
