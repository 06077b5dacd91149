set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp11
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 180 --timeout-method thread -m gpu -k "decode or detect or reorg or smoke" 2>&1 | tail -5
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/exp11/bench.json 2> gpurun_out/exp11/bench.err || { tail -20 gpurun_out/exp11/bench.err; exit 1; }
python3 -c "
import json
r=json.loads(open('gpurun_out/exp11/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'])
for k in ('post_processing','decode'):
    v=dict(r.get(k,{})); v.pop('what',None); print(k, v)
"
