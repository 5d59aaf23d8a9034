cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode resident --no-realistic "$@" 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d['device_bytes']//2**30)
except Exception as e: print('$*', 'failed', e)"; }
one --batch 256
one --batch 384
one --batch 512 --in-flight 8 --lf-streams 6
one --batch 128
