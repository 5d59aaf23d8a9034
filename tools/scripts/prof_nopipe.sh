#!/bin/bash
# kernel durations without overlap between batches (what each kernel costs on an otherwise idle GPU)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/nopipe
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/nopipe -o np -- python $R/bench.py --steps 10 --warmup 3 --no-pipeline --no-cpu-baseline > $R/gpurun_out/nopipe/bench.log 2>&1 < /dev/null
DB=$(find $R/gpurun_out/nopipe -name "*.db" | head -1); python $R/tools/rocpd_stats.py "$DB" 2>&1 | head -30
tail -c 600 $R/gpurun_out/nopipe/bench.log
