set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s12_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s12_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$TAG', d['value'], d['ms_per_step'], d['step_ms_median'], r['conv_stack_ms'])"; }
TAG=base b
export YOLO_HALO_CFG="13=61"; TAG=h13_61 b
export YOLO_HALO_CFG="13=70"; TAG=h13_70 b
export YOLO_HALO_CFG="13=70,26=70"; TAG=h13_26_70 b
export YOLO_HALO_CFG="104=61"; TAG=h104_61 b
unset YOLO_HALO_CFG
TAG=base2 b
export YOLO_IGEMM_CFG="3x104=82,3x52=82,3x26=82,3x13=82"; TAG=s2_82 b
export YOLO_IGEMM_CFG="1x13=84"; TAG=i13_84 b
export YOLO_IGEMM_CFG="1x13=97"; TAG=i13_97 b
