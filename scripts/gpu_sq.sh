# usage: bash scripts/gpu_sq.sh <tag> [bench flags]  -> gpurun_out/sq_<tag>/sq.txt (copy to profiles/)
set -e
tag=$1
shift
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
d=gpurun_out/sq_$tag
rm -rf $d; mkdir -p $d
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $d/p1 -o p -- $cmd > $d/p1.log 2>&1 || { tail -20 $d/p1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $d/p2 -o p -- $cmd > $d/p2.log 2>&1 || { tail -20 $d/p2.log; exit 1; }
python3 scripts/pmc_sq.py $d/p1 $d/p2 > $d/sq.txt
rm -rf $d/p1 $d/p2
cat $d/sq.txt
