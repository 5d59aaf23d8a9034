# same-box A/B: the tree's build vs jpegxl-rs_amd/lib_ab2/libjxl.so (a build of an earlier commit), headline streaming leg; extra arguments go to bench.py
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-realistic --mode streaming $@"
for i in 1 2 3; do
  one tree $B
  one ab2 env JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab2/libjxl.so $B
done
