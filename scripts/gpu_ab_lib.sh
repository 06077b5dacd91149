# usage: bash scripts/gpu_ab_lib.sh <rounds> <steps> <libA.so> <libB.so> [...]: interleaved A/B of whole library builds (paths relative to
# yolo_amd/lib/), one session, one box
set -e
cd "$GRAFT_REPO_ROOT"
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for l in "$@"; do
    export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/$l
    timeout -k 10 200 python3 bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null
    python3 -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'res_block' in x['kernel'] or 'front' in x['kernel']]; print('round $r $l: %.4f ms/step (min %.4f median %.4f)' % (d['ms_per_step'], d['step_ms_min'], d['step_ms_median']), [(x['kernel'][6:20], x['ms']) for x in k])"
  done
done
