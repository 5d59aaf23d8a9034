cd $GRAFT_REPO_ROOT
timeout 600 python tools/experiments/gpu_lf_conc.py 256 2 > gpurun_out/lfconc.log 2>&1
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras"
echo "=== prio 3" >> gpurun_out/lfconc.log
JXL_HIP_LF_PRIO=1 timeout 600 $B >> gpurun_out/lfconc.log 2>&1
