# more pixel workgroups beside an HF workgroup: IdctTileKernel<4> at 96 VGPRs (lib_ab2, -DJXL_IDCT_MINW=5: four wavefronts per SIMD beside the HF wavefront's 80 registers instead of three),
# plus the filter tile without its pad column (lib_ab3, + -DJXL_FPAD=0: 13.4 KB, seven workgroups beside a 68 KB HF workgroup); tree = compact alias tables only.  Same box, alternating, resident K = 30
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab3/libjxl.so timeout 900 python -m pytest tests -m gpu -x -q -k "vardct or fused or full_size or batch or shapes or strategy" 2>&1 | tail -2
one() { python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
for i in 1 2 3; do
  one tree
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab2/libjxl.so one idct96
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab3/libjxl.so one idct96_fpad0
done
one tree_nopipe --no-pipeline --steps 8
JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab2/libjxl.so one idct96_nopipe --no-pipeline --steps 8
JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab3/libjxl.so one idct96_fpad0_nopipe --no-pipeline --steps 8
