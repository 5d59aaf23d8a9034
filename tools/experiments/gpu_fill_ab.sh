# the pipeline fill of the driver's command (K = 20): capped (170 VGPRs, spills) against uncapped (252) one-wavefront-per-stream LF kernel for the first launches of a cold pipeline, and how many of them
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/sc
mkdir -p gpurun_out/fill_ab
for cfg in ${CFGS:-0:4 1:4 0:6 1:6 0:4 1:4}; do
  cap=${cfg%%:*}; wf=${cfg##*:}
  JXL_HIP_LF_WIDE_CAPPED=$cap timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-realistic --no-cpu-baseline --no-verify --wide-first $wf > gpurun_out/fill_ab/line_${cap}_${wf}.json 2> gpurun_out/fill_ab/err.log
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fill_ab/line_${cap}_${wf}.json").read().strip().splitlines()[-1])
    e = d["step_end_ms"]
    print("capped $cap wide_first $wf:", d["value"], "Mpx/s; steady", d["steady_state_ms_per_step"], "; step ends", [round(x) for x in e[:8]], "...", round(e[-1]))
except Exception as ex:
    print("capped $cap wide_first $wf: failed", ex)
PY
done | tee gpurun_out/fill_ab/result.txt
