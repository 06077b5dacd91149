set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_small_batch.py -x -q -s --timeout 120 --timeout-method thread 2>&1 | tail -60 > gpurun_out/s20_tests.txt
cat gpurun_out/s20_tests.txt
timeout -k 10 100 python3 -u scripts/b1_layers.py 1 split > gpurun_out/s20_b1_split.txt 2>&1
timeout -k 10 100 python3 -u scripts/b1_layers.py 1 > gpurun_out/s20_b1.txt 2>&1
