# Are the pixel kernels' in-pipeline times an occupancy effect?  Un-pipelined decode (every stage alone on the GPU), the fused filter / IDCT tile launches asking for more
# LDS than they use, so that as many workgroups fit a CU as beside an 80 KB HF workgroup (filters: 32 KB -> 5 per CU instead of 9-11; IDCT: 26.6 KB -> 6 instead of 8-12) and fewer.
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { label="$1"; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-pipeline --no-cpu-baseline --no-extras --distinct 32 --mode resident --no-realistic 2>gpurun_out/occ_err.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['ms_per_step'], d['stage_ms'])
except Exception as e: print('$label', 'failed', e)"; tail -n 2 gpurun_out/occ_err.log | grep -v amdgpu.ids | cut -c1-300; }
one base A=1
one filter_5wg JXL_HIP_PAD_LDS_FILTER=18700
one filter_4wg JXL_HIP_PAD_LDS_FILTER=26000
one filter_3wg JXL_HIP_PAD_LDS_FILTER=39000
one filter_7wg JXL_HIP_PAD_LDS_FILTER=8800
one idct_6wg JXL_HIP_PAD_LDS_IDCT=14000
one idct_4wg JXL_HIP_PAD_LDS_IDCT=27000
one idct_3wg JXL_HIP_PAD_LDS_IDCT=40000
one base A=1
