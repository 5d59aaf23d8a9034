# verdict r3 item 4's open experiment: confine the LF wavefronts (one-wavefront workgroups, ~650 in flight, on every SIMD of the chip) to a subset of
# the CUs with hipExtStreamCreateWithCUMask, optionally the tail's kernels to the rest; same box, alternating, resident, K = 30
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; timeout 400 env "$@" python bench.py $EXTRA --steps 30 --warmup 3 --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic 2>gpurun_out/cumask_err.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))
except Exception as e: print('$label', 'failed', e)"; tail -2 gpurun_out/cumask_err.log | cut -c1-300; }
for i in 1 2; do
  one base A=1
  one lf64 JXL_BENCH_CUMASK_LF=first:64
  one lf128 JXL_BENCH_CUMASK_LF=first:128
  one lf64_main192 JXL_BENCH_CUMASK_LF=first:64 JXL_BENCH_CUMASK_MAIN=not-first:64
done
one lf32 JXL_BENCH_CUMASK_LF=first:32
EXTRA="--main-tree-shape 1 --main-texture 1.0"
one cjxl_base A=1
one cjxl_lf128 JXL_BENCH_CUMASK_LF=first:128
one cjxl_lf64 JXL_BENCH_CUMASK_LF=first:64
