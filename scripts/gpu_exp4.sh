set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp4
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes 1 --layer-table gpurun_out/exp4/layers_lanes1.txt > gpurun_out/exp4/b1.json 2> gpurun_out/exp4/last.err
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/exp4/b3.json 2> gpurun_out/exp4/last.err
cat gpurun_out/exp4/layers_lanes1.txt
python3 -c "
import json
for f in ('b1','b3'):
    r=json.loads(open('gpurun_out/exp4/%s.json'%f).read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'], r['post_processing']['ms'])
"
: > gpurun_out/exp4/stamps.txt
for a in "12" "9" "15"; do timeout -k 10 60 python3 scripts/stamps.py --only $a >> gpurun_out/exp4/stamps.txt 2>&1; done
grep -v -E "amdgpu.ids" gpurun_out/exp4/stamps.txt
