set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp5
o=gpurun_out/exp5/bench.txt; : > $o
run() {  # label lanes
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $2 > gpurun_out/exp5/last.json 2> gpurun_out/exp5/last.err || { tail -5 gpurun_out/exp5/last.err; exit 1; }
  python3 -c "import sys,json; r=json.loads(open('gpurun_out/exp5/last.json').read().strip().splitlines()[-1]); print('$1 lanes=$2', r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'])" >> $o
}
for l in 3 4 5 6 8; do run default $l; done
export YOLO_HALO_CFG="13=64"; for l in 1 3 4; do run 13=64 $l; done
export YOLO_HALO_CFG="13=68"; for l in 1 3 4; do run 13=68 $l; done
export YOLO_HALO_CFG="13=68,104=69"; for l in 3; do run 13=68,104=69 $l; done
export YOLO_HALO_CFG="13=68,104=61"; for l in 3; do run 13=68,104=61 $l; done
export YOLO_HALO_CFG="13=68,26=68"; for l in 3; do run 13=68,26=68 $l; done
unset YOLO_HALO_CFG
run default 3
cat $o
