# kernel timeline of a cold cjxl-shaped pipeline (where do the 2.4 s before the first step go?)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/filltrace
cd /tmp && export TMPDIR=/tmp
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -o ft -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --no-verify --distinct 16 --no-realistic --cjxl-distinct 0 --main-tree-shape 1 --main-texture 5 --mode resident > /tmp/ft.log 2>&1 < /dev/null
cp $(find /tmp/ft -name '*kernel_trace.csv' | head -1) $R/gpurun_out/filltrace/kernel_trace.csv
ls -la $R/gpurun_out/filltrace/
grep -v rocprofv3 /tmp/ft.log | tail -5 | cut -c1-300
