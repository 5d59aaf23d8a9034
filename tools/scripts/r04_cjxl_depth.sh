# cjxl-shaped LF trees as the main workload: batches in flight x LF side streams (the LF stage is ~700 ms per launch there)
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 500 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --main-tree-shape 1 --main-texture 5 "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']; print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d['device_bytes']//2**30)
except Exception as e: print('$*', 'failed', e)"; }
one --mode resident --in-flight 11 --lf-streams 7
one --mode resident --in-flight 14 --lf-streams 10
one --mode resident --in-flight 16 --lf-streams 12
one --mode streaming --in-flight 11 --lf-streams 7
one --mode streaming --in-flight 14 --lf-streams 10
GPU_MAX_HW_QUEUES=24 one --mode streaming --in-flight 14 --lf-streams 10
