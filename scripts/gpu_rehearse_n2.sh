# N = 2 control flow of bench.py on a one-GPU box: two ranks share the GPU, gloo for the three scalar reductions
set -e
cd "$GRAFT_REPO_ROOT"
export YOLO_BENCH_BACKEND=gloo
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/rehearse_n2.json 2> gpurun_out/rehearse_n2.err || { tail -30 gpurun_out/rehearse_n2.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/rehearse_n2.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['global_batch'], d['sharded_512'])
"
