# kernel stats of the cjxl-shaped workload as the main workload (how long does the redo-only LfDecodeKernel pass take behind the SIMT launches?)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export JXL_BENCH_STREAM_CACHE=/tmp/jxl_streams
rm -rf /tmp/cs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cs -o cs -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --no-verify --distinct 16 --no-realistic --cjxl-distinct 0 --main-tree-shape 1 --main-texture 5 --mode resident > /tmp/cs.log 2>&1 < /dev/null
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/cs/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
import collections
d = collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v.sort()
    print('%-62s n %4d  total %9.1f  median %8.2f  p90 %8.2f  max %8.2f' % (k, len(v), sum(v), v[len(v)//2], v[int(len(v)*0.9)], v[-1]))
PY
grep '"metric"' /tmp/cs.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steady_state_ms_per_step'], d['stage_ms'])"
