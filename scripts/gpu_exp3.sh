set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp3
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "alternative_conv_tilings" --timeout 120 --timeout-method thread 2>&1 | tail -5
o=gpurun_out/exp3/layers.txt; : > $o
for cfg in -1 60 61 67 64 -1 60; do
  echo "== YOLO_CONV_CFG=$cfg" >> $o
  if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
  timeout -k 10 120 python3 scripts/bench_layer.py --only 6,9,12,15 --iters 30 2>/dev/null >> $o
done
cat $o
o=gpurun_out/exp3/bench.txt; : > $o
for cfg in -1 55 56 65 66 67 -1 56; do
  for lanes in 1 3; do
    echo "== YOLO_CONV_CFG=$cfg lanes=$lanes" >> $o
    if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $lanes > gpurun_out/exp3/last.json 2> gpurun_out/exp3/last.err || { tail -5 gpurun_out/exp3/last.err; exit 1; }
    python3 -c "import sys,json; r=json.loads(open('gpurun_out/exp3/last.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'])" >> $o
  done
done
unset YOLO_CONV_CFG
cat $o
: > gpurun_out/exp3/stamps.txt
for a in "12 --cfg 60" "12 --cfg 61" "9 --cfg 60" "15 --cfg 61"; do timeout -k 10 60 python3 scripts/stamps.py --only $a >> gpurun_out/exp3/stamps.txt 2>&1; done
cat gpurun_out/exp3/stamps.txt
