cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline --no-verify > gpurun_out/deep_$name.json 2> gpurun_out/deep_$name.err; }
run base JXL_BENCH_DEEP=0
run deep3 JXL_BENCH_DEEP=1
run deep4 JXL_BENCH_DEEP=1 JXL_BENCH_NBUF=4
run deep5 JXL_BENCH_DEEP=1 JXL_BENCH_NBUF=5
python - <<'PY'
import json
for b in ("base","deep3","deep4","deep5"):
    try:
        r=json.loads(open(f"gpurun_out/deep_{b}.json").read().strip().splitlines()[-1])
        print(b, r["value"], r["ms_per_step"], r["stage_ms"], round(r["device_bytes"]/2**30,1))
    except Exception as e: print(b, "ERR", e, open(f"gpurun_out/deep_{b}.err").read()[-800:])
PY
