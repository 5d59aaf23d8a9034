R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03a
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03a/stats -o lfconc -- python $R/tools/experiments/gpu_lf_conc.py 256 2 > $R/gpurun_out/r03a/lfconc.log 2>&1 < /dev/null
find $R/gpurun_out/r03a -name "*kernel_stats.csv" | head; cat $(find $R/gpurun_out/r03a -name "*kernel_stats.csv" | head -1) | head -20
find $R/gpurun_out/r03a -name "*kernel_trace.csv" -size +20M -delete
