set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "alternative_conv_tilings and (90 or 91 or 92 or 93 or 88)" 2>&1 | tail -5
for c in 88 90 91 92 93; do
  echo "== cfg $c"
  YOLO_CONV_CFG=$c timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 64 --only 5,8,11,14,18,21 2>&1 | grep -v amdgpu.ids | grep -v weighted || exit 1
done
