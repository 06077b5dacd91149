set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp10
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_shard.py tests/test_preprocess.py tests/test_abi.py -x -q -s --timeout 180 --timeout-method thread -m gpu 2>&1 | tail -15
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python3 bench.py > gpurun_out/exp10/bench.json 2> gpurun_out/exp10/bench.err || { tail -20 gpurun_out/exp10/bench.err; exit 1; }
python3 -c "
import json
r=json.loads(open('gpurun_out/exp10/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'])
for k in ('roofline','post_processing','decode','sharded_512','fp32','cpu_baseline'):
    v=dict(r.get(k,{})); v.pop('kernels',None); v.pop('what',None); v.pop('measured',None); v.pop('sample',None); print(k, v)
"
