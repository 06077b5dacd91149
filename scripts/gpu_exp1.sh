# round-2 experiment 1: 256x128 two-workgroup halo tile (cfg 50/51/52) against the shipped tiling, per layer shape
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp1
for cfg in -1 50 51 52 -1 50; do
  echo "== YOLO_CONV_CFG=$cfg" >> gpurun_out/exp1/layers.txt
  if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
  timeout -k 10 120 python3 scripts/bench_layer.py --only 6,9,12,15 --iters 30 >> gpurun_out/exp1/layers.txt 2>&1
done
unset YOLO_CONV_CFG
cat gpurun_out/exp1/layers.txt
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "alternative_conv_tilings or conv_layer_bf16" --timeout 120 --timeout-method thread 2>&1 | tail -5
