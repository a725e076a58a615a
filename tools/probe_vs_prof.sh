set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/tools/fuse_probe.py --reps 10 --only feather:f32 --kernels > $R/gpurun_out/pp_plain.txt 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_prof -o x -- python3 $R/tools/fuse_probe.py --reps 10 --only feather:f32 --kernels > $R/gpurun_out/pp_prof.txt 2>&1
timeout -k 10 120 python3 $R/tools/fuse_probe.py --reps 10 --only feather:f32 --kernels > $R/gpurun_out/pp_plain2.txt 2>&1
rocm-smi --showclocks > $R/gpurun_out/pp_clocks.txt 2>&1 || true
