cd $GRAFT_REPO_ROOT
for p in 0 2 4 6; do JXL_HIP_LF_PRIO=$p timeout 300 python tools/experiments/gpu_corun_lf.py 2>&1 | grep "in flight"; done > gpurun_out/corun_lf.log 2>&1
LANE=2 JXL_HIP_LF_PRIO=0 timeout 300 python tools/experiments/gpu_corun_lf.py 2>&1 | grep "in flight" >> gpurun_out/corun_lf.log
cat gpurun_out/corun_lf.log
