# bench.jxl (the reference's criterion input: Modular, weighted predictor, 6643-node tree, 128 clusters): where does ModularGroupFastKernel spend its cycles?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/bj
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bj/stats -o s -- python $R/tools/experiments/gpu_benchjxl.py 1 > $R/gpurun_out/bj/stats.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/bj/p1 -o a -- python $R/tools/experiments/gpu_benchjxl.py 1 > $R/gpurun_out/bj/p1.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/bj/p2 -o b -- python $R/tools/experiments/gpu_benchjxl.py 1 > $R/gpurun_out/bj/p2.log 2>&1 < /dev/null
python $R/tools/pmc_sum.py $R/gpurun_out/bj/p1 | cut -c1-300
python $R/tools/pmc_sum.py $R/gpurun_out/bj/p2 | cut -c1-300
head -8 $(find $R/gpurun_out/bj/stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
find $R/gpurun_out/bj -name "*kernel_trace.csv" -delete; find $R/gpurun_out/bj -name "*counter_collection.csv" -delete
