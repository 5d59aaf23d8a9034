# kernel timeline of a few pipelined steps (rocprofv3 kernel trace + tools/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr && mkdir -p /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-verify > /tmp/tr/bench.log 2>&1 < /dev/null
python $R/tools/trace_gaps.py $(find /tmp/tr -name "*kernel_trace.csv" | head -1) | tail -45
