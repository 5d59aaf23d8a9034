# HF stage: streams per wavefront (JXL_HIP_HF_LANES) inside the pipeline — fewer, denser wavefronts leave SIMDs to the pixel kernels
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode resident "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']; r=d.get('workload_realistic') or d['config'].get('workload_realistic') or {}
print('HF_LANES=$JXL_HIP_HF_LANES', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], '| realistic', r.get('value'), r.get('ms_per_step'), r.get('stage_ms'))"; }
for l in 34 45 64 27 23 17; do JXL_HIP_HF_LANES=$l one; done
