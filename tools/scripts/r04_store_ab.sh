# fused filter kernel: dword-packed u8 RGB stores (tree) vs three byte stores per pixel (lib_ab, -DJXL_NO_PACKED_STORE) vs packed + non-temporal (lib_ab2, -DJXL_PACKED_STORE_NT)
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
timeout 900 python -m pytest tests -m gpu -x -q -k "vardct or fused or full_size or batch or shapes" 2>&1 | tail -3
one() { python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
for i in 1 2 3; do
  one packed
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab/libjxl.so one bytes
  JXL_HIP_LIBJXL=$GRAFT_REPO_ROOT/jpegxl-rs_amd/lib_ab2/libjxl.so one packed_nt
done
