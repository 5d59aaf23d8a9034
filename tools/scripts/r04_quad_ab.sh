# weighted-predictor LF streams: four lanes per stream (quad) vs one lane per stream; LF stage alone and inside the pipeline
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cjxl or weighted or wp or corrupted_round4" 2>&1 | tail -3
one() { env $1 timeout 500 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --distinct 16 --no-realistic --cjxl-distinct 0 --main-tree-shape 1 --main-texture 5 --mode resident "${@:2}" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))
except Exception as e: print('$*', 'failed', e)"; }
for i in 1 2; do one X=1; one JXL_HIP_LF_NOQUAD=1; done
one X=1 --lane-stride-lf 16
one X=1 --lane-stride-lf 4
SHAPE=1 timeout 200 python tools/experiments/gpu_single_breakdown.py 2>&1 | grep "rep 2"
JXL_HIP_LF_NOQUAD=1 SHAPE=1 timeout 200 python tools/experiments/gpu_single_breakdown.py 2>&1 | grep "rep 2"
