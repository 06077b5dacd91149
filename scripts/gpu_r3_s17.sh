set -e
cd "$GRAFT_REPO_ROOT"
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --layer-table gpurun_out/s17_layers.txt "$@" > gpurun_out/s17_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s17_b.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['step_ms_median'])" && sed -n 3p gpurun_out/s17_layers.txt; }
TAG=base b
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_exp11.so
TAG=exp11 b
unset YOLO_HIP_LIB
TAG=base b
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_exp11.so
TAG=exp11 b
