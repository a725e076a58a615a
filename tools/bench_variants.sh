set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-side"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bvp1 -o x -- $B > $R/gpurun_out/bv_p1.json 2>/dev/null
timeout -k 10 200 $B > $R/gpurun_out/bv_q.json 2>/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bvp2 -o x -- $B > $R/gpurun_out/bv_p2.json 2>/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bvp3 -o x -- $B > $R/gpurun_out/bv_p3.json 2>/dev/null
