# headline (streaming, no extras) under variants of the LF kernel's LDS use; usage: r04_bench_variants.sh
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { "${@:2}" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-realistic --mode streaming 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])"; }
for i in 1 2; do
  one tree env

  one round3 env JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so
done
