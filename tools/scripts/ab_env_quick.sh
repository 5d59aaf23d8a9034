# A/B of one build under settings of an environment variable, resident pipeline only (fast): bash tools/scripts/ab_env_quick.sh NAME VALUE_A VALUE_B [more bench args]
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
N=$1; A=$2; B=$3; shift 3
one() { env $N=$1 timeout 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic --no-verify "${@:2}" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])
except Exception as e: print('$N=$1 failed', e)"; }
for i in 1 2; do one $A "$@"; one $B "$@"; done
