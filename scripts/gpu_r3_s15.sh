set -e
cd "$GRAFT_REPO_ROOT"
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s15_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s15_b.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['step_ms_median'])"; }
TAG=base b
export YOLO_FRONT_TH=4; TAG=th4 b; TAG=th4_l1 b --lanes 1
unset YOLO_FRONT_TH
TAG=base_l1 b --lanes 1
TAG=base b
