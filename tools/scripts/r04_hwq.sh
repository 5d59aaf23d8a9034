# GPU_MAX_HW_QUEUES 16 (bench.py's default) against 24 and 32 on the reported lines; one box, alternating
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 700 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], 'resident', d['resident_mpixel_per_s'], 'realistic', c['workload_realistic']['value'], 'cjxl', c['workload_cjxl_shape']['value'], d['stage_ms'])
except Exception as e: print('$label', 'failed', e)"; }
for i in 1 2; do
  GPU_MAX_HW_QUEUES=16 one k20_q16 --gpus 1 --steps 20 --warmup 5
  GPU_MAX_HW_QUEUES=24 one k20_q24 --gpus 1 --steps 20 --warmup 5
done
GPU_MAX_HW_QUEUES=24 one k100_q24
