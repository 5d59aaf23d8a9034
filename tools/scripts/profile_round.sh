set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02a
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02a/stats -o r02a -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/r02a/bench.log 2>$R/gpurun_out/r02a/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02a/fetch -o f -- python $R/bench.py --batch 256 --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/r02a/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02a/write -o w -- python $R/bench.py --batch 256 --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/r02a/write.log 2>&1
find $R/gpurun_out/r02a -name "*.csv" | head -20
tail -c 600 $R/gpurun_out/r02a/bench.log
# keep only what is needed (64 MiB limit): drop the big kernel traces
find $R/gpurun_out/r02a -name "*kernel_trace.csv" -size +20M -delete
du -sh $R/gpurun_out/r02a
