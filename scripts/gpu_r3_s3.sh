set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  rm -rf /tmp/kt$n
  YOLO_BENCH_RING_MB=6000 YOLO_BENCH_STREAMS=$n timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$n -o kt -- python3 $GRAFT_REPO_ROOT/scripts/bench_layer.py --iters 24 --only 12 > /tmp/kt$n.log 2>&1 || { tail -5 /tmp/kt$n.log; exit 1; }
  find /tmp/kt$n -type f | head
  f=$(find /tmp/kt$n -name "*kernel_trace*" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kt_streams$n.csv
done
