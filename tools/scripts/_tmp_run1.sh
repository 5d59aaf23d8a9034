cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/selfclean.json 2> gpurun_out/selfclean.err < /dev/null
python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/selfclean.json").read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["stage_ms"], r.get("verified_vs_oracle"))
except Exception as e: print("ERR", e, open("gpurun_out/selfclean.err").read()[-1500:])
PY
