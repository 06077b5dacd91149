# parity of the conv kernels + the bench at three lanes and one lane, twice (A/B runs of kernel changes)
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -1 gpurun_out/ab_tests.log
for l in 0 1 0 1; do
  if [ $l = 1 ]; then extra="--lanes 1"; else extra=""; fi
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $extra > gpurun_out/ab.json
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$extra', d['value'], d['ms_per_step'], r['achieved'], r['conv_stack_ms'])"
done
