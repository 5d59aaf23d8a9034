cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras"
for cfg in "JXL_BENCH_DEEP=1 A=--in-flight=8" "JXL_BENCH_DEEP=0 A=--in-flight=8" "JXL_BENCH_DEEP=1 A=--in-flight=10" "JXL_BENCH_DEEP=1 A=--lane-stride-lf=8" ; do
  echo "=== $cfg" >> gpurun_out/b1.log
  env $cfg bash -c "timeout 600 $B \$A" >> gpurun_out/b1.log 2>&1
done
echo "=== legacy 3 in flight" >> gpurun_out/b1.log
JXL_BENCH_DEEP=0 timeout 600 $B --in-flight 3 --lf-streams 2 --lane-stride-lf 64 >> gpurun_out/b1.log 2>&1
