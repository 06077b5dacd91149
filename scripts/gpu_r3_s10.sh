set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 -u scripts/check_cfg.py --cfgs 91,96,97 2>&1 | grep -v amdgpu.ids
for c in 91 96 97; do
  echo "== cfg $c"
  YOLO_CONV_CFG=$c timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 64 --only 5,8,11,14,18,21 2>&1 | grep -v amdgpu.ids | grep -v weighted || exit 1
done
for c in 88 91 97; do
YOLO_IGEMM_CFG="1x104=$c,1x52=$c,1x26=$c" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s10_b.json && python3 -c "import json; d=json.loads(open('gpurun_out/s10_b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('e2e $c', d['value'], d['ms_per_step'], r['achieved'], r['conv_stack_ms'])"
done
