set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp8
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "register_epilogue or alternative_conv_tilings" --timeout 120 --timeout-method thread 2>&1 | tail -5
o=gpurun_out/exp8/layers.txt; : > $o
for cfg in -1 75 88 -1; do
  echo "== YOLO_CONV_CFG=$cfg" >> $o
  if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
  timeout -k 10 120 python3 scripts/bench_layer.py --only 6,9,12,15,10,13,8,11,14 --iters 30 2>/dev/null >> $o
done
unset YOLO_CONV_CFG
cat $o
