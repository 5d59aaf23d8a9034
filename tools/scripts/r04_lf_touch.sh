cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
for i in 1 2; do
  echo "== touch"; python tools/experiments/gpu_lf_wp_time.py 256 16 8 2>&1 | grep "tree shape"
  echo "== no touch"; JXL_HIP_LF_PRIO=16 python tools/experiments/gpu_lf_wp_time.py 256 16 8 2>&1 | grep "tree shape"
done
one() { "${@:2}" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-realistic --mode streaming 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
for i in 1 2; do
  one touch env
  one notouch env JXL_HIP_LF_PRIO=16
done
