# --tail-streams 2 (the filter stage on its own stream beside the next batch's IDCT, two sets of pixel planes) again after the occupancy work; + HF lanes per wavefront; resident K = 40, one box, alternating
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 500 env "$@" python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 32 --no-realistic --cjxl-distinct 0 --mode resident $EXTRA 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d['device_bytes']>>30)
except Exception as e: print('$label', 'failed', e)"; }
for i in 1 2; do
  EXTRA=""; one base A=1
  EXTRA="--tail-streams 2"; one tail2 A=1
done
EXTRA=""; one hf_lanes45 JXL_HIP_HF_LANES=45
EXTRA=""; one hf_lanes27 JXL_HIP_HF_LANES=27
EXTRA="--tail-streams 2 --main-texture 1.0"; one tail2_realistic A=1
EXTRA="--main-texture 1.0"; one base_realistic A=1
