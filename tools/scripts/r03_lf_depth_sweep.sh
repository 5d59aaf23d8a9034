# is the steady state bound by the LF stages in flight?  (in flight, LF side streams, hardware queues), resident mode
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode ${MODE:-resident} --no-realistic "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']
print('Q=$GPU_MAX_HW_QUEUES $*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], [round(b-a) for a,b in zip([0]+s,s)][:14])"; }
one --in-flight 10 --lf-streams 6
one --in-flight 10 --lf-streams 9
one --in-flight 12 --lf-streams 8
one --in-flight 14 --lf-streams 10
GPU_MAX_HW_QUEUES=24 one --in-flight 14 --lf-streams 12
GPU_MAX_HW_QUEUES=24 one --in-flight 10 --lf-streams 9
one --in-flight 8 --lf-streams 6
one --in-flight 7 --lf-streams 6
