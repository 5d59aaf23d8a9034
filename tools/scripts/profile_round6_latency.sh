# The latency passes of profile_round6.sh alone (after a kernel of that path changed): one 4K frame, one stream per wavefront — kernel stats and SQ instruction counts —, and bench.jxl.
set -x
TAG=${TAG:-r06c}
R=$GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for T in 0 1; do
  LF_STRIDE=64 HF_LPW=1 TREE_SHAPE=$T timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/lat${T}_stats -o p -- python $R/tools/experiments/one_batch_decode.py 4k 1 4 > $R/gpurun_out/$TAG/lat${T}_stats.log 2>&1 < /dev/null
  LF_STRIDE=64 HF_LPW=1 TREE_SHAPE=$T timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/lat${T}_SQ -o p -- python $R/tools/experiments/one_batch_decode.py 4k 1 4 > $R/gpurun_out/$TAG/lat${T}_SQ.log 2>&1 < /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/benchjxl_stats -o p -- python $R/tools/experiments/gpu_r6_benchjxl.py > $R/gpurun_out/$TAG/benchjxl_stats.log 2>&1 < /dev/null
find $R/gpurun_out/$TAG -name "*agent_info.csv" -delete
du -sh $R/gpurun_out/$TAG
