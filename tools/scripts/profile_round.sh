# Profiling passes of a round (run under gpurun): kernel trace + stats of the default bench (streaming + resident + realistic), then one PMC
# pass each for FETCH_SIZE and WRITE_SIZE on a single 256-frame decode, and a kernel timeline of the pipelined steps.
# Usage: bash tools/scripts/profile_round.sh   (tag below)
set -x
TAG=${TAG:-r03a}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --distinct 64 --realistic-distinct 32 > $R/gpurun_out/$TAG/bench.log 2>$R/gpurun_out/$TAG/bench.err < /dev/null
python $R/tools/trace_gaps.py $(find $R/gpurun_out/$TAG/stats -name "*kernel_trace.csv" | head -1) 9 3 > $R/gpurun_out/$TAG/timeline.txt 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/fetch -o f -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --mode resident --distinct 32 --no-realistic --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/$TAG/fetch.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/write -o w -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --mode resident --distinct 32 --no-realistic --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/$TAG/write.log 2>&1 < /dev/null
find $R/gpurun_out/$TAG -name "*.csv" | head -20
tail -c 600 $R/gpurun_out/$TAG/bench.log
# keep only what is needed (64 MiB limit): drop the big kernel traces
find $R/gpurun_out/$TAG -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/$TAG -name "*counter_collection.csv" -size +30M -delete
du -sh $R/gpurun_out/$TAG
