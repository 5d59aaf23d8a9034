# pipeline depth again after the occupancy work (compact alias tables, IDCT at 96 VGPRs): (batches in flight, LF streams) for the headline frames and for the cjxl-shaped ones; resident, K = 60 / 40, one box
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 500 python bench.py --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --mode resident "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])
except Exception as e: print('$*', 'failed', e)"; }
for d in "9 5" "10 6" "11 7" "12 8"; do set -- $d; one --steps 60 --in-flight $1 --lf-streams $2; done
one --steps 60 --in-flight 11 --lf-streams 7
for d in "11 7" "12 8" "13 9" "14 10"; do set -- $d; one --steps 40 --main-tree-shape 1 --main-texture 1.0 --wp-in-flight $1 --wp-lf-streams $2 --in-flight $1 --lf-streams $2; done
