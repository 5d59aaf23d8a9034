#!/bin/bash
# round 5: tools/api_concurrent (the Rust crate's call sequence from T pthreads) on the bench frames
set -e
cd "$(dirname "$0")/../.."
export JXL_BENCH_STREAM_CACHE=${JXL_BENCH_STREAM_CACHE:-/tmp/sc}
N=${N_DISTINCT:-64}
python - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
streams = bench.make_streams($N, 3840, 2160, 1)
os.makedirs("/tmp/jxl_frames", exist_ok=True)
for i, s in enumerate(streams):
    open(f"/tmp/jxl_frames/f{i:03d}.jxl", "wb").write(s)
PY
GPU_MAX_HW_QUEUES=16 tools/_build/api_concurrent jpegxl-rs_amd/lib/libjxl.so /tmp/jxl_frames ${THREADS:-1,8,64} ${PER_THREAD:-10} 3
