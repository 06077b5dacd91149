set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s16_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s16_b.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['step_ms_median'])"; }
TAG=base b
export YOLO_IGEMM_CFG="3x104=89,3x52=89,3x26=89,3x13=89"; TAG=s2_89 b
export YOLO_IGEMM_CFG="3x104=89,3x52=89,3x26=89,3x13=89,1x13=89"; TAG=s2_89_13 b
unset YOLO_IGEMM_CFG
TAG=base b
export YOLO_IGEMM_CFG="3x104=89,3x52=89,3x26=89,3x13=89"; TAG=s2_89 b
for c in 88 89; do YOLO_CONV_CFG=$c timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --only 4,7,10,13,14 2>&1 | grep -v amdgpu.ids | grep -v weighted; done
