# HF lanes per wavefront on the realistic (1.85 bpp) frames, where the HF stage (75 ms) is the longest stage of a step; resident K = 40, one box
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 500 env "$@" python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --mode resident --main-texture ${TEX:-1.0} 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])
except Exception as e: print('$label', 'failed', e)"; }
one lanes34 A=1
one lanes27 JXL_HIP_HF_LANES=27
one lanes23 JXL_HIP_HF_LANES=23
one lanes34 A=1
one lanes17 JXL_HIP_HF_LANES=17
