# --tail-streams 2 on the lines that are reported: the full bench process (streaming headline + realistic + cjxl-shaped legs) at the driver's K = 20 and at K = 100, against the default; one box, alternating
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 700 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], 'resident', d['resident_mpixel_per_s'], 'realistic', c['workload_realistic']['value'], 'cjxl', c['workload_cjxl_shape']['value'], d['stage_ms'], d.get('verified_vs_oracle'), d['device_bytes']>>30)
except Exception as e: print('$label', 'failed', e)"; }
for i in 1 2; do
  one k20_base --gpus 1 --steps 20 --warmup 5
  one k20_tail2 --gpus 1 --steps 20 --warmup 5 --tail-streams 2
done
one k100_tail2 --tail-streams 2
one k100_base
