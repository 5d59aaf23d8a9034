cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
JXL_HIP_TIME_PREPARE=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify --mode streaming --no-realistic --distinct 64 2>&1 | grep -E "Prepare of|metric" | cut -c1-330 | tail -24
