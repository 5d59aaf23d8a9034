# pipeline fill of the cjxl-shaped workload at the driver's K = 20: when do the steps complete?
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --main-tree-shape 1 --main-texture 5 --mode streaming "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']; print('$*', d['value'], d['ms_per_step'], [round(b-a) for a,b in zip([0]+s,s)], d['stage_ms']['lf'])
except Exception as e: print('$*', 'failed', e)"; }
one --in-flight 14 --lf-streams 10
one --in-flight 11 --lf-streams 7
one --in-flight 8 --lf-streams 7
one --in-flight 14 --lf-streams 10 --lane-stride-lf 16
one --in-flight 14 --lf-streams 10 --lane-stride-lf 4
