# usage: bash scripts/gpu_profile.sh <tag> [extra bench.py flags, e.g. --lanes 1]   -> gpurun_out/prof_<tag>/{stats.txt, traffic.txt}
set -e
tag=$1
shift
extra="$*"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
d=gpurun_out/prof_$tag
rm -rf $d; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra > $d/bench_under_prof.log 2>&1 || { tail -30 $d/bench_under_prof.log; exit 1; }
python3 scripts/prof_summary.py $(ls $d/kt/*.db | head -1) $d/stats.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra" > /dev/null
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/pmc_fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $extra > $d/pmc_fetch.log 2>&1 || { tail -20 $d/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/pmc_write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $extra > $d/pmc_write.log 2>&1 || { tail -20 $d/pmc_write.log; exit 1; }
python3 scripts/pmc_traffic.py $d/pmc_fetch $d/pmc_write > $d/traffic.txt
rm -rf $d/kt/*.db $d/pmc_fetch/*counter_collection.csv.bak 2>/dev/null || true
head -20 $d/stats.txt; cat $d/traffic.txt
