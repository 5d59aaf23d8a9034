# streaming pipeline fill: how many cold LF stages take the wide kernel, how many prepare workers
cd $GRAFT_REPO_ROOT
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --mode ${MODE:-streaming} --no-realistic "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_end_ms']
print('$*', d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d.get('streaming',{}).get('prepare_ms_per_batch'), [round(b-a) for a,b in zip([0]+s,s)])"; }
one --wide-first 3 --prepare-threads 3
one --wide-first 4 --prepare-threads 3
one --wide-first 4 --prepare-threads 4
one --wide-first 5 --prepare-threads 3
JXL_BENCH_LF_PRIO=first one --wide-first 4 --prepare-threads 4
JXL_BENCH_LF_PRIO=first one --wide-first 6 --prepare-threads 3
one --wide-first 4 --prepare-threads 4 --parse-threads 16
MODE=resident one --wide-first 3
MODE=resident one --wide-first 4
MODE=resident one --wide-first 5
