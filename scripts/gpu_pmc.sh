# usage: bash scripts/gpu_pmc.sh "<layer idx list>" tag cfg "<counters pass1>" ["<counters pass2>" ...]
set -e
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
layers=$1; tag=$2; export YOLO_CONV_CFG=$3; shift 3
i=0
for ctrs in "$@"; do
  i=$((i+1))
  mkdir -p gpurun_out/pmc_${tag}_$i
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 scripts/bench_layer.py --only $layers --iters 3 > gpurun_out/pmc_${tag}_$i/log.txt 2>&1 || { tail -20 gpurun_out/pmc_${tag}_$i/log.txt; exit 1; }
done
ls gpurun_out | grep pmc_${tag}
