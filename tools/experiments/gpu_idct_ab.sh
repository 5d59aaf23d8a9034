# A/B of two builds of the library, stage times alone (HIP events; nothing else on the GPU): lib_v/libjxl_old.so against lib/libjxl.so
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/sc
mkdir -p gpurun_out/idct_ab
for tex in 0 5; do
for v in old new old new; do
  lib=jpegxl-rs_amd/lib/libjxl.so; [ $v = old ] && lib=jpegxl-rs_amd/lib_v/libjxl_old.so
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/experiments/gpu_stage_alone.py 256 6 $tex 2>&1 | tail -1 | cut -c1-400
done; done | tee gpurun_out/idct_ab/result.txt
