# kernel timeline of a few pipelined steps (rocprofv3 kernel trace + tools/trace_gaps.py); extra bench arguments / environment pass through
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${TRACE_OUT:-$R/gpurun_out/trace.txt}
rm -rf /tmp/tr && mkdir -p /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 14 --warmup 2 --no-cpu-baseline --no-extras --no-verify "$@" > /tmp/tr/bench.log 2>&1 < /dev/null
tail -c 700 /tmp/tr/bench.log > $OUT
python $R/tools/trace_gaps.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) 9 2 >> $OUT
