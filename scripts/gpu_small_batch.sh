# end-to-end latency at small batch (the reference's own published figure is a batch-1 latency)
set -e
cd "$GRAFT_REPO_ROOT"
for b in 1 2 4 8 16 32; do
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --batch $b > gpurun_out/sb.json
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/sb.json').read().strip().splitlines()[-1]); print('batch $b: %.3f ms per step, %.1f img/s, conv stack %.3f ms' % (d['ms_per_step'], d['value'], d['roofline']['conv_stack_ms']))"
done
