cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --distinct 64 --mode streaming"
run() { echo "=== $*" >> gpurun_out/b1.log; env "$@" bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1; }
run X=1 A="--wide-first 3"
run X=1 A="--wide-first 4"
run X=1 A="--wide-first 6"
run X=1 A="--wide-first 4 --prepare-threads 4"
