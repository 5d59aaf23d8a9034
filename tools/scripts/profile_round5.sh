# Profiling passes of round 5 (run under gpurun): kernel trace + stats of the driver's bench command (library pipeline), then one PMC pass each for FETCH_SIZE and
# WRITE_SIZE on plain batch decodes of the three workloads (256 4K frames; 8 HDR 8K frames, EPF 3, f32; 2 Modular 8K frames), and a kernel timeline of the pipelined steps.
# Separate --pmc passes, no trace domains beside --kernel-trace (MI355X_MICROARCH.md: HBM / rocprofv3 section).
set -x
TAG=${TAG:-r05a}
R=$GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
[ "${SKIP_STATS:-0}" = 1 ] || timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --distinct 64 --realistic-distinct 32 > $R/gpurun_out/$TAG/bench.log 2>$R/gpurun_out/$TAG/bench.err < /dev/null
[ "${SKIP_STATS:-0}" = 1 ] || python $R/tools/trace_gaps.py $(find $R/gpurun_out/$TAG/stats -name "*kernel_trace.csv" | head -1) 9 3 > $R/gpurun_out/$TAG/timeline.txt 2>&1
for W in "4k 256 2" "hdr8k 8 2" "mod8k 2 2"; do
  set -- $W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/${1}_$C -o p -- python $R/tools/experiments/one_batch_decode.py $1 $2 $3 > $R/gpurun_out/$TAG/${1}_$C.log 2>&1 < /dev/null
  done
done
# VALU instruction counts of the 4K kernels (SQ_INSTS_VALU: the bench line's valu_issue)
timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/4k_SQ -o p -- python $R/tools/experiments/one_batch_decode.py 4k 256 2 > $R/gpurun_out/$TAG/4k_SQ.log 2>&1 < /dev/null
find $R/gpurun_out/$TAG -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/$TAG -name "*counter_collection.csv" -size +30M -delete
find $R/gpurun_out/$TAG -name "*agent_info.csv" -delete
du -sh $R/gpurun_out/$TAG
tail -c 400 $R/gpurun_out/$TAG/bench.log
