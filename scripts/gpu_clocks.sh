# samples clocks / power while the bench runs (is the step power-limited?)
cd "$GRAFT_REPO_ROOT"
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clocks.txt &
SM=$!
timeout -k 10 200 python3 bench.py --steps 2000 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/clk_bench.json
wait $SM
cat gpurun_out/clocks.txt | cut -c1-300 | sed -n '1,40p'
python3 -c "import json; d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
