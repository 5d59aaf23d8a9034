# the SIMT LF wavefronts at s_setprio 3 (JXL_HIP_LF_PRIO=1) inside the pipeline: does the LF stage get closer to its 215 ms alone, and can the pipeline then run with fewer LF stages in flight?  resident, one box, alternating
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 500 env "$@" python bench.py --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --mode resident $EXTRA 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])
except Exception as e: print('$label', 'failed', e)"; }
EXTRA="--steps 40"
for i in 1 2; do
  one base A=1
  one lfprio JXL_HIP_LF_PRIO=1
done
EXTRA="--steps 40 --in-flight 10 --lf-streams 6"; one lfprio_10_6 JXL_HIP_LF_PRIO=1
EXTRA="--steps 40 --in-flight 9 --lf-streams 5"; one lfprio_9_5 JXL_HIP_LF_PRIO=1
EXTRA="--steps 40 --main-tree-shape 1 --main-texture 1.0"; one cjxl_base A=1; one cjxl_lfprio JXL_HIP_LF_PRIO=1
