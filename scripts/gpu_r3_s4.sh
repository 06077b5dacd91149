set -e
cd "$GRAFT_REPO_ROOT"
for b in 12 24 36 48 64 72 96 144; do
  echo "batch $b"
  timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch $b --only 12,9,15 2>&1 | grep -v amdgpu.ids | grep -v weighted
done
