#!/bin/bash
# round 6: kernel timeline of the FILL of the timed region (an idle pipeline, K fresh jobs): where do the ~450 ms until the first step is done go?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06/fill
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06/fill -o f -- python $R/bench.py --gpus 1 --steps ${K:-20} --warmup ${WU:-5} --no-realistic --no-extras --no-cpu-baseline ${EXTRA_ARGS:-} > $R/gpurun_out/r06/fill/bench.log 2>&1 < /dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r06/fill/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("jxlhip::", "").replace("void ", "")[:34], r.get("Queue_Id", "")) for r in rows)
# the timed region = the last burst: find the largest idle gap before the last 6 HF launches
hf = [k for k in ks if k[2].startswith("HfDecodeSimt")]
t_first_hf = hf[-${K:-20}][0]
# start of the region: the first kernel after the longest gap preceding t_first_hf
prev_end = 0; start = ks[0][0]; best = 0
for s, e, n, q in ks:
    if s > t_first_hf: break
    if prev_end and s - prev_end > best and s > t_first_hf - 1.2e9: best = s - prev_end; start = s
    prev_end = max(prev_end, e)
print("timed region starts", (t_first_hf - start) / 1e6, "ms before its first HF launch; idle gap before it", best / 1e6, "ms")
for s, e, n, q in ks:
    if s < start or s > start + 700e6: continue
    if (e - s) < 1.0e6: continue
    print(f"{(s - start) / 1e6:9.2f} -> {(e - start) / 1e6:9.2f}  ({(e - s) / 1e6:7.2f} ms)  q{q}  {n}")
PY
tail -c 300 $R/gpurun_out/r06/fill/bench.log
find $R/gpurun_out/r06/fill -name "*.csv" -size +5M -delete
