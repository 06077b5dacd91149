set -e
for c in $1; do echo "== cfg $c"; YOLO_CONV_CFG=$c timeout -k 10 200 python -u scripts/bench_layer.py --only $2 2>&1 | grep -v amdgpu.ids; done
