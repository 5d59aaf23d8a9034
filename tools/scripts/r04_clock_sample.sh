# engine clock / power while the pipeline runs (is the step's 0.60 of VALU issue a clock that sits below 2.4 GHz?): rocm-smi sampled twice a second beside a resident K = 80 run
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
( for i in $(seq 1 160); do echo "t=$i $(rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i 'sclk\|Average Graphics Package Power\|Socket Graphics\|GPU use' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.5; done ) > gpurun_out/clock_samples.log &
SAMP=$!
timeout 400 python bench.py --steps 80 --warmup 3 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])"
kill $SAMP 2>/dev/null
cat gpurun_out/clock_samples.log | tail -120
