set -e
cd "$GRAFT_REPO_ROOT"
for e in "" exp7 exp8 exp9; do
  if [ -n "$e" ]; then export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/yolo_amd/lib/libyolo_hip_$e.so; fi
  echo "== lib ${e:-shipped} cfg 91"
  YOLO_CONV_CFG=91 timeout -k 10 100 python3 -u scripts/bench_layer.py --iters 40 --batch 64 --only 5,8,11,14 2>&1 | grep -v amdgpu.ids | grep -v weighted || exit 1
done
