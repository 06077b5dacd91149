set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_small_batch.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
mkdir -p gpurun_out/b1t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1t -o split -- python3 scripts/b1_trace.py 1 > gpurun_out/b1t/log.txt 2>&1
f=$(find gpurun_out/b1t -name '*kernel_trace.csv' | head -1)
python3 scripts/b1_trace.py --read "$f" > gpurun_out/s21_b1_trace.txt
rm -rf gpurun_out/b1t
tail -3 gpurun_out/s21_b1_trace.txt
