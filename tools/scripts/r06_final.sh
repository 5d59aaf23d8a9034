# final pass of round 6 on one box: GPU suite, the driver's command (K = 20) and the default bench line (K = 100), each timed
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r06d}
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
mkdir -p gpurun_out/$TAG
[ "${SKIP_SUITE:-0}" = 1 ] || (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4; echo "pytest-exit $?") > gpurun_out/$TAG/gpu_suite.txt 2>&1
t0=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_line_k20.json 2> gpurun_out/$TAG/bench_err.log; t1=$(date +%s)
[ "${SKIP_K100:-0}" = 1 ] || timeout 1500 python bench.py > gpurun_out/$TAG/bench_line.json 2>> gpurun_out/$TAG/bench_err.log; t2=$(date +%s)
echo "bench K=20: $((t1 - t0)) s wall, default bench: $((t2 - t1)) s wall" > gpurun_out/$TAG/bench_wall.txt
cat gpurun_out/$TAG/gpu_suite.txt gpurun_out/$TAG/bench_wall.txt; cut -c1-300 gpurun_out/$TAG/bench_line_k20.json; cut -c1-300 gpurun_out/$TAG/bench_line.json; tail -5 gpurun_out/$TAG/bench_err.log
