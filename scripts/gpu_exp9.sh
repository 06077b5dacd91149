set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp9
o=gpurun_out/exp9/bench.txt; : > $o
run() {  # label lanes
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $2 > gpurun_out/exp9/last.json 2> gpurun_out/exp9/last.err || { tail -5 gpurun_out/exp9/last.err; exit 1; }
  python3 -c "import sys,json; r=json.loads(open('gpurun_out/exp9/last.json').read().strip().splitlines()[-1]); print('$1 lanes=$2', r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'])" >> $o
}
IG="1x52=88,1x26=88,1x13=88,3x104=88,3x52=88,3x26=88,3x13=88"
for l in 1 3; do run default $l; done
export YOLO_IGEMM_CFG=$IG; for l in 1 3; do run ig88 $l; done
export YOLO_HALO_CFG="13=73,26=72"; for l in 1 3; do run ig88+13=73,26=72 $l; done
export YOLO_HALO_CFG="13=73,26=72,52=72"; for l in 1 3; do run ig88+13=73,26=72,52=72 $l; done
export YOLO_HALO_CFG="13=73,26=72,52=72,104=74"; for l in 1 3; do run ig88+all $l; done
unset YOLO_HALO_CFG YOLO_IGEMM_CFG
for l in 1 3; do run default $l; done
cat $o
