cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras"
run() { echo "=== $*" >> gpurun_out/b1.log; env "$@" bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1; }
run JXL_BENCH_LDS_BUDGET=0 A="--in-flight 9 --hf-streams 2"
run JXL_BENCH_LDS_BUDGET=0 A="--in-flight 8 --hf-streams 3"
run JXL_BENCH_LDS_BUDGET=0 JXL_HIP_HF_PRIO=0 A="--in-flight 8 --hf-streams 3"
run JXL_BENCH_LDS_BUDGET=0 JXL_HIP_HF_LANES=64 A="--in-flight 8 --hf-streams 3"
run JXL_BENCH_LDS_BUDGET=20000 A="--in-flight 9 --hf-streams 2"
