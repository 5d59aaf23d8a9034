# final pass of round 5 on one box: GPU suite, the driver's command (K = 20) and the default bench line (K = 100)
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05f}
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4; echo "pytest-exit $?") > gpurun_out/$TAG/gpu_suite.txt 2>&1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_line_k20.json 2> gpurun_out/$TAG/bench_err.log
timeout 1500 python bench.py --no-extras > gpurun_out/$TAG/bench_line.json 2>> gpurun_out/$TAG/bench_err.log
cat gpurun_out/$TAG/gpu_suite.txt; cut -c1-300 gpurun_out/$TAG/bench_line_k20.json; cut -c1-300 gpurun_out/$TAG/bench_line.json
