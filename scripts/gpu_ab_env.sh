# usage: bash scripts/gpu_ab_env.sh <rounds> <steps> <VAR> <value> [<value> ...]: interleaved A/B of the default against
# VAR=<value> settings of the XP library (YOLO_HALO_CFG / YOLO_IGEMM_CFG maps), one session, one box
set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_xp.so
rounds=$1; steps=$2; var=$3; shift 3
for r in $(seq 1 $rounds); do
  for c in default "$@"; do
    if [ "$c" = default ]; then unset $var; else export $var=$c; fi
    timeout -k 10 200 python3 bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null
    python3 -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('round $r $var=$c: %.4f ms/step (min %.4f median %.4f)' % (d['ms_per_step'], d['step_ms_min'], d['step_ms_median']))"
  done
done
