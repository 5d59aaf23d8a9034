# Profiling passes of round 6 (run under gpurun): as profile_round5.sh — kernel trace + stats of the driver's bench command, one PMC pass each for FETCH_SIZE and WRITE_SIZE on
# plain batch decodes of the three workloads — plus the LATENCY path this round rebuilt: one 4K frame (gradient LF tree and cjxl-shaped tree) through the wave-wide LF
# decoders and the wave-wide HF kernel (kernel stats, SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES), and bench.jxl (big-tree form).
# Separate --pmc passes, no trace domains beside --kernel-trace (MI355X_MICROARCH.md: HBM / rocprofv3 section).
set -x
TAG=${TAG:-r06a}
R=$GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
[ "${SKIP_STATS:-0}" = 1 ] || timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --distinct 64 --realistic-distinct 32 > $R/gpurun_out/$TAG/bench.log 2>&1 < /dev/null
[ "${SKIP_STATS:-0}" = 1 ] || python $R/tools/trace_gaps.py $(find $R/gpurun_out/$TAG/stats -name "*kernel_trace.csv" | head -1) 9 3 > $R/gpurun_out/$TAG/timeline.txt 2>&1
for W in "4k 256 2" "hdr8k 8 2" "mod8k 2 2"; do
  set -- $W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/${1}_$C -o p -- python $R/tools/experiments/one_batch_decode.py $1 $2 $3 > $R/gpurun_out/$TAG/${1}_$C.log 2>&1 < /dev/null
  done
done
timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/4k_SQ -o p -- python $R/tools/experiments/one_batch_decode.py 4k 256 2 > $R/gpurun_out/$TAG/4k_SQ.log 2>&1 < /dev/null
# ---- latency path: one frame, one stream per wavefront
for T in 0 1; do
  LF_STRIDE=64 HF_LPW=1 TREE_SHAPE=$T timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/lat${T}_stats -o p -- python $R/tools/experiments/one_batch_decode.py 4k 1 4 > $R/gpurun_out/$TAG/lat${T}_stats.log 2>&1 < /dev/null
  LF_STRIDE=64 HF_LPW=1 TREE_SHAPE=$T timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/lat${T}_SQ -o p -- python $R/tools/experiments/one_batch_decode.py 4k 1 4 > $R/gpurun_out/$TAG/lat${T}_SQ.log 2>&1 < /dev/null
done
for C in FETCH_SIZE WRITE_SIZE; do
  LF_STRIDE=64 HF_LPW=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/lat0_$C -o p -- python $R/tools/experiments/one_batch_decode.py 4k 1 4 > $R/gpurun_out/$TAG/lat0_$C.log 2>&1 < /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/benchjxl_stats -o p -- python $R/tools/experiments/gpu_r6_benchjxl.py > $R/gpurun_out/$TAG/benchjxl_stats.log 2>&1 < /dev/null
find $R/gpurun_out/$TAG -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/$TAG -name "*counter_collection.csv" -size +30M -delete
find $R/gpurun_out/$TAG -name "*agent_info.csv" -delete
du -sh $R/gpurun_out/$TAG
tail -c 400 $R/gpurun_out/$TAG/bench.log
