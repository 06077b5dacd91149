set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp7
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "alternative_conv_tilings or conv_layer or forward_bf16 or front_fusion" --timeout 120 --timeout-method thread 2>&1 | tail -5
o=gpurun_out/exp7/layers.txt; : > $o
for cfg in -1 70 71 -1; do
  echo "== YOLO_CONV_CFG=$cfg" >> $o
  if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
  timeout -k 10 120 python3 scripts/bench_layer.py --only 9,12,15 --iters 30 2>/dev/null >> $o
done
unset YOLO_CONV_CFG
cat $o
o=gpurun_out/exp7/bench.txt; : > $o
run() {  # label lanes
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $2 > gpurun_out/exp7/last.json 2> gpurun_out/exp7/last.err || { tail -5 gpurun_out/exp7/last.err; exit 1; }
  python3 -c "import sys,json; r=json.loads(open('gpurun_out/exp7/last.json').read().strip().splitlines()[-1]); print('$1 lanes=$2', r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'])" >> $o
}
for l in 1 3; do run default $l; done
export YOLO_HALO_CFG="26=70"; for l in 1 3; do run 26=70 $l; done
export YOLO_HALO_CFG="26=70,52=70"; for l in 1 3; do run 26=70,52=70 $l; done
export YOLO_HALO_CFG="26=71"; for l in 1 3; do run 26=71 $l; done
unset YOLO_HALO_CFG
for l in 1 3; do run default $l; done
cat $o
