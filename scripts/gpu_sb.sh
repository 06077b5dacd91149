# small-batch plan: tests + latency table + kernel trace of one batch-1 forward
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_small_batch.py -x -q --timeout 120 --timeout-method thread > gpurun_out/sb_tests.txt 2>&1 || { tail -30 gpurun_out/sb_tests.txt; exit 1; }
tail -2 gpurun_out/sb_tests.txt
timeout -k 10 300 python3 -u scripts/small_batch.py 1 2 4 8 > gpurun_out/sb_lat.txt 2>&1
cat gpurun_out/sb_lat.txt
mkdir -p gpurun_out/b1t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1t -o split -- python3 scripts/b1_trace.py 1 > gpurun_out/b1t/log.txt 2>&1
f=$(find gpurun_out/b1t -name '*kernel_trace.csv' | head -1)
python3 scripts/b1_trace.py --read "$f" > gpurun_out/sb_b1_trace.txt
rm -rf gpurun_out/b1t
tail -1 gpurun_out/sb_b1_trace.txt
