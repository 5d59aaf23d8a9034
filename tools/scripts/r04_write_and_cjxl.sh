# (1) the WRITE_SIZE pass of the round's PMC pair again (it hung in rocprofv3's finalisation inside profile_round.sh), (2) the cjxl-shaped leg inside the full bench process at other pipeline depths
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04f
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04f/write -o w -- python $R/bench.py --batch 256 --steps 1 --warmup 1 --no-pipeline --mode resident --distinct 32 --no-realistic --no-cpu-baseline --no-extras --no-verify > $R/gpurun_out/r04f/write.log 2>&1 < /dev/null
echo "write pass exit $?"; find $R/gpurun_out/r04f/write -name "*.csv" | head
find $R/gpurun_out/r04f -name "*kernel_trace.csv" -size +20M -delete
cd $R
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
one() { timeout 600 python bench.py --no-cpu-baseline --no-extras --no-verify --mode streaming --realistic-distinct 16 --distinct 64 "$@" 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['workload_cjxl_shape']; print('$*', 'main', d['value'], d['ms_per_step'], 'cjxl', c['value'], c['ms_per_step'], c['steady_state_ms_per_step'], c['stage_ms'])
except Exception as e: print('$*', 'failed', e)"; }
one --steps 20 --warmup 5 --wp-in-flight 11 --wp-lf-streams 7
one --steps 20 --warmup 5 --wp-in-flight 13 --wp-lf-streams 9
one --steps 20 --warmup 5 --wp-in-flight 12 --wp-lf-streams 8
one --steps 100 --wp-in-flight 11 --wp-lf-streams 7
one --steps 100 --wp-in-flight 13 --wp-lf-streams 9
