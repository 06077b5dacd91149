set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/b1t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1t -o def -- python3 scripts/b1_trace.py 1 default > gpurun_out/b1t/log.txt 2>&1
f=$(find gpurun_out/b1t -name '*kernel_trace.csv' | head -1)
python3 scripts/b1_trace.py --read "$f" > gpurun_out/s22_b1_trace_default.txt
rm -rf gpurun_out/b1t
tail -3 gpurun_out/s22_b1_trace_default.txt
