set -e
cd "$GRAFT_REPO_ROOT"
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s13_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s13_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$TAG', d['value'], d['ms_per_step'], d['step_ms_median'], r['conv_stack_ms'])"; }
TAG=shipped b
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
TAG=xp b
unset YOLO_HIP_LIB
TAG=shipped b
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
TAG=xp b
rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2
