# Profiling passes of a round (run under gpurun): kernel trace + stats of the default bench, then one PMC pass each for FETCH_SIZE and
# WRITE_SIZE on a single 256-frame decode.  Usage: bash tools/scripts/profile_round.sh   (tag below)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02c
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02c/stats -o r02c -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/r02c/bench.log 2>$R/gpurun_out/r02c/bench.err < /dev/null
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/fetch -o f -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/r02c/fetch.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/write -o w -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/r02c/write.log 2>&1 < /dev/null
find $R/gpurun_out/r02c -name "*.csv" | head -20
tail -c 600 $R/gpurun_out/r02c/bench.log
# keep only what is needed (64 MiB limit): drop the big kernel traces
find $R/gpurun_out/r02c -name "*kernel_trace.csv" -size +20M -delete
du -sh $R/gpurun_out/r02c
