# A/B of two builds of libjxl.so on one box: the tree's build vs jpegxl-rs_amd/lib_ab/libjxl.so (a variant built by hand), alternating
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
for i in 1 2; do
  one tree "$@"
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so one variant "$@"
done
