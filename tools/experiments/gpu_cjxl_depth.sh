# cjxl-shaped LF trees (weighted predictor): LF stage 591 ms per launch against steps of ~84 ms = seven LF streams all busy.  Does a deeper pipeline (more LF launches in flight) lift it?
cd $GRAFT_REPO_ROOT
TAG=${TAG:-cjxl_depth}
export JXL_BENCH_STREAM_CACHE=/tmp/sc
mkdir -p gpurun_out/$TAG
COMMON="--steps ${STEPS:-20} --warmup 5 --main-tree-shape 1 --main-texture 5 --no-extras --no-realistic --no-cpu-baseline --no-verify"
for cfg in ${CFGS:-11:7 14:10 16:12 20:14}; do
  inf=${cfg%%:*}; lfs=${cfg##*:}
  timeout 600 python bench.py $COMMON --in-flight $inf --lf-streams $lfs > gpurun_out/$TAG/line_${inf}_${lfs}.json 2> gpurun_out/$TAG/err_${inf}_${lfs}.log
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$TAG/line_${inf}_${lfs}.json").read().strip().splitlines()[-1])
    print("in_flight $inf lf_streams $lfs:", d["value"], "Mpx/s", d["ms_per_step"], "ms/step", "steady", d.get("steady_state_ms_per_step"), d.get("stage_ms"))
except Exception as e:
    print("in_flight $inf lf_streams $lfs: failed", e)
PY
done
