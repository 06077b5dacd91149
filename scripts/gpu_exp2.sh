set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp2
o=gpurun_out/exp2/bench.txt
: > $o
for cfg in -1 53 54 -1 54; do
  for lanes in 1 3; do
    echo "== YOLO_CONV_CFG=$cfg lanes=$lanes" >> $o
    if [ "$cfg" = "-1" ]; then unset YOLO_CONV_CFG; else export YOLO_CONV_CFG=$cfg; fi
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $lanes > gpurun_out/exp2/last.json 2> gpurun_out/exp2/last.err || { tail -5 gpurun_out/exp2/last.err; exit 1; }
    python3 -c "import sys,json; r=json.loads(open('gpurun_out/exp2/last.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['roofline']['conv_stack_ms'])" >> $o
  done
done
unset YOLO_CONV_CFG
cat $o
for a in "12" "12 --cfg 50" "12 --cfg 51" "15" "15 --cfg 51" "9" "9 --cfg 50"; do timeout -k 10 60 python3 scripts/stamps.py --only $a >> gpurun_out/exp2/stamps.txt 2>&1; done
cat gpurun_out/exp2/stamps.txt
timeout -k 10 200 python3 -u -m pytest tests/wip_no_alloc.py tests/test_abi.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
