# SIMT HF kernel: compact alias tables in LDS (6 bytes per slot: 4-byte slot + u16 frequency per symbol; tree) vs the 8-byte slots (lib_ab, -DJXL_HF_WIDE_ALIAS);
# GPU suite first, then same box, alternating, resident K = 30 (+ the realistic and cjxl-shaped frames as the main workload)
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
one() { python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'), d['device_bytes']>>20)"; }
for i in 1 2 3; do
  one compact
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so one wide
done
one compact_realistic --main-texture 1.0
JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so one wide_realistic --main-texture 1.0
one compact_cjxl --main-texture 1.0 --main-tree-shape 1
JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so one wide_cjxl --main-texture 1.0 --main-tree-shape 1
JXL_HIP_DEBUG_LDS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic 2>&1 | grep "LDS sizing" | sort | uniq -c | head -5
