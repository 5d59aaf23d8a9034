# the cjxl-shaped workload as the headline leg under pipeline / lane variants
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'], d['step_end_ms'][:3], d.get('verified_vs_oracle'))"; }
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-realistic --mode streaming --main-tree-shape 1 --main-texture 5 --distinct 32"
one base $B
one lanes16 $B --lane-stride-lf 4
one lanes32 $B --lane-stride-lf 2
one inflight14 $B --in-flight 14
one inflight14_lanes16 $B --in-flight 14 --lane-stride-lf 4
