set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf /tmp/kt3
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /tmp/kt3.log 2>&1 || { tail -5 /tmp/kt3.log; exit 1; }
cp /tmp/kt3/kt_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/kt_lanes3.csv
tail -c 300 /tmp/kt3.log
