# batches in flight x LF side streams x cold wide LF stages at the driver's K = 20, streaming (six LF streams leave an LF stage of ~390 ms no slack at 65 ms per step)
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode $MODE --no-realistic "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']
print('$MODE $*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], [round(b-a) for a,b in zip([0]+s,s)])"; }
MODE=streaming
for i in 1 2 3; do
one --in-flight 10 --lf-streams 6 --wide-first 3
one --in-flight 10 --lf-streams 6 --wide-first 4
one --in-flight 11 --lf-streams 7 --wide-first 3
one --in-flight 11 --lf-streams 7 --wide-first 4
one --in-flight 11 --lf-streams 7 --wide-first 5
done
