# the filter stage on its own stream beside the IDCT of the next batch (two sets of pixel planes) vs one tail stream; same box, alternating
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic "$@" 2>gpurun_out/tail_err.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'), d['device_bytes']//2**30)
except Exception as e: print('$*', 'failed', e)"; tail -3 gpurun_out/tail_err.log | cut -c1-300; }
for i in 1 2; do
  one --tail-streams 1
  one --tail-streams 2
done
one --tail-streams 2 --hf-streams 2
