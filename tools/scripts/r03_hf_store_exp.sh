# what do the HF stage's scattered coefficient stores cost the pixel kernels beside it?  (JXL_HIP_HF_PRIO: 1 normal, 5 no stores, 13 sequential stores; wrong pixels)
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode resident --no-realistic "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HF_PRIO=$JXL_HIP_HF_PRIO NOCOEF=$JXL_HIP_IDCT_NOCOEF', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])"; }
JXL_HIP_HF_PRIO=1 one
JXL_HIP_HF_PRIO=5 one
JXL_HIP_HF_PRIO=13 one
JXL_HIP_HF_PRIO=5 JXL_HIP_IDCT_NOCOEF=1 one
JXL_HIP_HF_PRIO=13 JXL_HIP_IDCT_NOCOEF=1 one
JXL_HIP_HF_PRIO=1 JXL_HIP_IDCT_NOCOEF=1 one
