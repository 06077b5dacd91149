set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u bench.py --steps 10 --warmup 3 --h2d --layer-table gpurun_out/layers_bf16.txt > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
