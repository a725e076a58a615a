#!/bin/bash
# Run on the GPU box: the work of every rank of an 8-way split of the default workload, one after the other on ONE
# GPU (bench.py --as-rank R --of 8: no process group, no collective), one JSON line per rank under gpurun_out/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for r in 0 1 2 3 4 5 6 7; do
  timeout -k 10 200 python3 bench.py --as-rank $r --of 8 --cpu-pairs 0 > gpurun_out/rank_${r}_of_8.json 2> gpurun_out/rank_of_8.err
done
python3 - <<PY
import json
rows = [json.loads(open(f"gpurun_out/rank_{r}_of_8.json").read().strip().splitlines()[-1]) for r in range(8)]
out = {"what": "bench.py --as-rank R --of 8, R = 0..7, one after the other on one MI355X: each rank's share of the c3 workload "
               "without the collective (projection of the 8-GPU strong-scaling step; not a multi-GPU measurement)",
       "ranks": [dict(r["as_rank"], registration_graph_replay_ms=(r.get("registration_graph_replay") or {}).get("ms_per_step"),
                      kernels={k: v["ms_per_step"] for k, v in r["kernels"].items()}) for r in rows]}
out["slowest_registration_graph_replay_ms"] = max((x["registration_graph_replay_ms"] or 0.0) for x in out["ranks"])
out["slowest_registration_ms"] = max(r["as_rank"]["registration_ms"] for r in rows)
out["slowest_fusion_ms"] = max(r["as_rank"]["fusion_ms"] for r in rows)
json.dump(out, open("gpurun_out/rank_of_8.json", "w"), indent=1)
print(out["slowest_registration_ms"], out["slowest_fusion_ms"], [r["as_rank"]["registration_ms"] for r in rows], [r["as_rank"]["fusion_ms"] for r in rows])
PY
