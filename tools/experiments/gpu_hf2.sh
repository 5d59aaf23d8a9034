# two HF stages in flight beside the tail (three coefficient sets) against one: headline + textured + cjxl-shaped frames at the driver's K = 20
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/sc
mkdir -p gpurun_out/hf2
for h in ${HFS:-1 2 1 2}; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-verify --hf-streams $h > gpurun_out/hf2/line_$h.json 2> gpurun_out/hf2/err.log
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/hf2/line_$h.json").read().strip().splitlines()[-1]); c = d["config"]
    print("hf_streams $h:", d["value"], d["steady_state_ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms"].items()})
    for k in ("workload_realistic", "workload_cjxl_shape"): print("    ", k, c[k]["value"], c[k]["steady_state_ms_per_step"], {kk: round(v, 1) for kk, v in c[k]["stage_ms"].items()})
except Exception as ex:
    print("hf_streams $h: failed", ex)
PY
done | tee gpurun_out/hf2/result.txt
