set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_exp10.so
b() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras "$@" > gpurun_out/s14_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s14_b.json').read().strip().splitlines()[-1]); print('double $YOLO_DOUBLE_CLASS $1 $2', d['ms_per_step'], d['step_ms_median'])"; }
for m in 0 1 2 4 8 16 32 64 128 256 512 1024 2048 0; do export YOLO_DOUBLE_CLASS=$m; b; done
