cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --distinct 64 --mode streaming --no-realistic"
run() { echo "=== $*" >> gpurun_out/b1.log; env "$@" bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1; }
run X=1 A="--prepare-threads 6"
run X=1 A="--prepare-threads 6 --wide-first 4"
run X=1 A="--prepare-threads 6 --parse-threads 4"
run X=1 A="--prepare-threads 8 --parse-threads 4 --wide-first 4"
