# Where do the wave cycles of the pixel kernels and the HF stage go?  SQ counters (two passes) on one un-pipelined decode of 256 frames.
R=$GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
mkdir -p $R/gpurun_out/sq
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $R/gpurun_out/sq/sq_counters.txt
CMD="python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --mode resident --distinct 32 --no-realistic --no-cpu-baseline --no-extras --no-verify"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/sq/p1 -o a -- $CMD > $R/gpurun_out/sq/p1.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/sq/p2 -o b -- $CMD > $R/gpurun_out/sq/p2.log 2>&1 < /dev/null
python $R/tools/pmc_sum.py $R/gpurun_out/sq/p1 > $R/gpurun_out/sq/p1_sum.csv
python $R/tools/pmc_sum.py $R/gpurun_out/sq/p2 > $R/gpurun_out/sq/p2_sum.csv
find $R/gpurun_out/sq -name "*kernel_trace.csv" -delete; find $R/gpurun_out/sq -name "*counter_collection.csv" -size +5M -delete
cat $R/gpurun_out/sq/p1_sum.csv | cut -c1-300; cat $R/gpurun_out/sq/p2_sum.csv | cut -c1-300; tail -3 $R/gpurun_out/sq/p1.log $R/gpurun_out/sq/p2.log
