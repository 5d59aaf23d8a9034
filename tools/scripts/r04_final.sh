# final pass of a round on one box: GPU suite, the default bench line (K = 100) and the driver's command (K = 20), then the profiling passes (tools/scripts/profile_round.sh)
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04f}
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4; echo "pytest-exit $?") > gpurun_out/$TAG/gpu_suite.txt 2>&1
timeout 900 python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench_err.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_line_k20.json 2>> gpurun_out/$TAG/bench_err.log
TAG=$TAG bash tools/scripts/profile_round.sh > gpurun_out/$TAG/profile_round.log 2>&1
cat gpurun_out/$TAG/gpu_suite.txt; cut -c1-400 gpurun_out/$TAG/bench_line.json; cut -c1-400 gpurun_out/$TAG/bench_line_k20.json; tail -n 5 gpurun_out/$TAG/profile_round.log
