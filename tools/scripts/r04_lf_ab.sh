# LF stage alone: round-3 build (jpegxl-rs_amd/lib_ab/libjxl.so) vs the tree's, gradient tree; then the tree's build on the cjxl-shaped tree
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
for i in 1 2; do
  echo "== tree build"; python tools/experiments/gpu_lf_wp_time.py 256 16 64,8 2>&1 | grep "tree shape"
  echo "== round-3 build"; JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so JXL_AB_SHAPE0_ONLY=1 python tools/experiments/gpu_lf_wp_time.py 256 16 64,8 2>&1 | grep "tree shape"
done
