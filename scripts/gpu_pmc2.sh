# usage: bash scripts/gpu_pmc2.sh "<layer idx list>" tag  -> gpurun_out/pmc2_<tag>_{1,2}
set -e
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
layers=$1; tag=$2
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  d=gpurun_out/pmc2_${tag}_$i
  rm -rf $d; mkdir -p $d
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $d -o p -- python3 scripts/bench_layer.py --only $layers --iters 3 > $d/log.txt 2>&1 || { tail -20 $d/log.txt; exit 1; }
done
python3 scripts/pmc_show.py "gpurun_out/pmc2_${tag}_*" halo
