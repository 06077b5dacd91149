set -e
cd "$GRAFT_REPO_ROOT"
for e in "" exp1 exp2 exp3; do
  if [ -n "$e" ]; then export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_$e.so; fi
  echo "== lib ${e:-shipped}"
  timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 64 --only 12,9,15 2>&1 | grep -v amdgpu.ids | grep -v weighted
  timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 48 --only 12 2>&1 | grep -v amdgpu.ids | grep -v weighted
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s5_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s5_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('e2e', d['value'], d['ms_per_step'], r['achieved'], r['conv_stack_ms'])"
done
