cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic --no-verify"
run() { echo "=== $*" >> gpurun_out/b1.log; env "$@" bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1; }
run X=1 A=""
run JXL_HIP_IDCT_NOCOEF=1 A=""
run X=1 A="--no-pipeline"
run JXL_HIP_IDCT_NOCOEF=1 A="--no-pipeline"
