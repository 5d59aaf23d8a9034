cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-verify --mode streaming --realistic-distinct 16 --distinct 64 "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['workload_cjxl_shape']; print('$*', 'main', d['value'], d['ms_per_step'], 'cjxl', c['value'], c['ms_per_step'], c['steady_state_ms_per_step'], c['stage_ms'])
except Exception as e: print('$*', 'failed', e)"; }
one --steps 100 --wp-in-flight 11 --wp-lf-streams 7
one --steps 100 --wp-in-flight 13 --wp-lf-streams 9
one --steps 20 --warmup 5 --wp-in-flight 11 --wp-lf-streams 7
one --steps 20 --warmup 5 --wp-in-flight 13 --wp-lf-streams 9
