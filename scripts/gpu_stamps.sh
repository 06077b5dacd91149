set -e
mkdir -p gpurun_out
: > gpurun_out/stamps.txt
for spec in "$@"; do
  idx=${spec%%:*}; cfg=${spec##*:}
  timeout -k 10 100 python -u scripts/stamps.py --only $idx --cfg $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/stamps.txt
done
cat gpurun_out/stamps.txt
