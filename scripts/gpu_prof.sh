set -e
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/bench_under_prof.log 2>&1 || { tail -30 gpurun_out/prof/bench_under_prof.log; exit 1; }
ls -R gpurun_out/prof | head -30
