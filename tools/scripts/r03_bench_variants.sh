cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras"
run() { echo "=== $*" >> gpurun_out/b1.log; env "$@" bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1; }
run X=1 A=""
run X=1 A="--lane-stride-lf 16 --in-flight 8 --lf-streams 7"
run X=1 A="--lane-stride-lf 4"
