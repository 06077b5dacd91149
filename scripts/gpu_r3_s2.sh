# saturated (multi-stream) vs serial per-layer rates + new ingest tests
set -e
cd "$GRAFT_REPO_ROOT"
for n in 1 4; do
  echo "== streams $n"
  YOLO_BENCH_RING_MB=6000 YOLO_BENCH_STREAMS=$n timeout -k 10 300 python3 -u scripts/bench_layer.py --iters 60 --only 4,5,6,7,8,9,10,11,12,13,14,15,18,21 2>&1 | grep -v amdgpu.ids
done
