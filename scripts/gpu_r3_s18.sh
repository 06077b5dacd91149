set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "front or forward or lanes or detect_one_call" 2>&1 | tail -3
b() { timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --layer-table gpurun_out/s18_layers.txt "$@" > gpurun_out/s18_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s18_b.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['step_ms_median'])" && sed -n 3p gpurun_out/s18_layers.txt; }
TAG=new b
TAG=new_l1 b --lanes 1
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_old.so
TAG=old b
TAG=old_l1 b --lanes 1
unset YOLO_HIP_LIB
TAG=new b
