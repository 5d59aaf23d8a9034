# streaming pipeline fill: how many cold LF stages take the wide kernel, how many prepare workers
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode streaming --no-realistic "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']
print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['streaming']['prepare_ms_per_batch'], [round(b-a) for a,b in zip([0]+s,s)])"; }
one --wide-first 3 --prepare-threads 3
one --wide-first 6 --prepare-threads 3
one --wide-first 6 --prepare-threads 6
one --wide-first 4 --prepare-threads 6
one --wide-first 8 --prepare-threads 4
one --wide-first 6 --prepare-threads 4 --parse-threads 12
one --wide-first 12 --prepare-threads 6
