# usage: bash scripts/gpu_ab_cfg.sh <rounds> <steps> <cfg> [<cfg> ...]: interleaved A/B of the default tile choice against forced
# YOLO_CONV_CFG values (a forced value only applies to the layers it is valid for), all inside one session on one box
set -e
cd "$GRAFT_REPO_ROOT"
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for c in default "$@"; do
    if [ "$c" = default ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$c; fi
    timeout -k 10 200 python3 bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null
    python3 -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('round $r cfg $c: %.4f ms/step (min %.4f median %.4f)' % (d['ms_per_step'], d['step_ms_min'], d['step_ms_median']))"
  done
done
