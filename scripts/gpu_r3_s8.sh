set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 -u scripts/check_cfg.py 2>&1 | grep -v amdgpu.ids
for c in 88 90 91 92 93 94 95; do
  echo "== cfg $c"
  YOLO_CONV_CFG=$c timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 64 --only 5,8,11,14,18,21 2>&1 | grep -v amdgpu.ids | grep -v weighted || exit 1
done
