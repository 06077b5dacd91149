# usage: bash scripts/gpu_profile.sh <tag> [extra bench.py flags, e.g. --lanes 1]   -> gpurun_out/prof_<tag>/{stats.txt, traffic.txt, traffic_table.json}
# (copy stats.txt / traffic.txt to profiles/<round>_*.txt and traffic_table.json to profiles/traffic_table.json: bench.py reads
#  the table and refuses it when the kernel sources have changed since)
set -e
tag=$1
shift
extra="$*"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
d=gpurun_out/prof_$tag
rm -rf $d; mkdir -p $d
cmd="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d/kt -o kt -- $cmd > $d/bench_under_prof.log 2>&1 || { tail -30 $d/bench_under_prof.log; exit 1; }
python3 scripts/prof_summary.py $(ls $d/kt/*.db | head -1) $d/stats.txt "rocprofv3 --kernel-trace --stats -- $cmd" > /dev/null
cmd2="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $extra"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/pmc_fetch -o p -- $cmd2 > $d/pmc_fetch.log 2>&1 || { tail -20 $d/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/pmc_write -o p -- $cmd2 > $d/pmc_write.log 2>&1 || { tail -20 $d/pmc_write.log; exit 1; }
python3 scripts/pmc_traffic.py $d/pmc_fetch $d/pmc_write $d/traffic_table.json 64 416 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -- $cmd2" > $d/traffic.txt
rm -rf $d/kt/*.db $d/pmc_fetch $d/pmc_write 2>/dev/null || true
head -20 $d/stats.txt; cat $d/traffic.txt
