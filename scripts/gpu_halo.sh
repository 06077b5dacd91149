set -e
for c in $1; do
  echo "== correctness cfg $c"
  YOLO_CONV_CFG=$c timeout -k 10 120 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_layer or forward_fp32_vs_reference or forward_fp32_rect or forward_bf16" --timeout 60 --timeout-method thread 2>&1 | tail -4
done
for c in $1; do echo "== cfg $c"; YOLO_CONV_CFG=$c timeout -k 10 100 python -u scripts/bench_layer.py --only 6,9,12,15 2>&1 | grep -v amdgpu.ids; done
