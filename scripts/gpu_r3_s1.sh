# round-3 session 1: sanity tests, baseline bench, env-var tile experiments, batches-in-flight experiment
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/s1_tests.log 2>&1 || { tail -30 gpurun_out/s1_tests.log; exit 1; }
tail -1 gpurun_out/s1_tests.log
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s1_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s1_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$TAG', d['value'], d['ms_per_step'], r['achieved'], r['conv_stack_ms'])"; }
TAG=base3 b
TAG=base1 b --lanes 1
TAG=base3b b
export YOLO_HALO_CFG="13=61"; TAG=h13_61 b
export YOLO_HALO_CFG="13=70"; TAG=h13_70 b
export YOLO_HALO_CFG="26=70"; TAG=h26_70 b
export YOLO_HALO_CFG="26=70,13=70"; TAG=h26_13_70 b
unset YOLO_HALO_CFG
timeout -k 10 300 python3 -u scripts/exp_inflight.py --steps 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s1_inflight.txt
