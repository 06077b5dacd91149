set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
b() { timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" > gpurun_out/s19_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s19_b.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['step_ms_median'], d['step_ms_min'])"; }
for rep in 1 2 3; do
unset YOLO_IGEMM_CFG YOLO_HALO_CFG
TAG=base b
export YOLO_IGEMM_CFG="3x104=89,3x52=89,3x26=89,3x13=89"; TAG=s2_89 b
export YOLO_IGEMM_CFG="1x13=97"; TAG=i13_97 b
unset YOLO_IGEMM_CFG
export YOLO_HALO_CFG="13=70"; TAG=h13_70 b
done
