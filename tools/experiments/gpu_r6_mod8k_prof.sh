#!/bin/bash
# round 6: kernel stats + timeline of BASELINE config 4 jobs (tools/experiments/gpu_mod8k_jobs.py ${CFG:-8:2})
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06/mod8k
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/mod8k -o m -- python $R/tools/experiments/gpu_mod8k_jobs.py ${CFG:-8:2} > $R/gpurun_out/r06/mod8k/log.txt 2>&1 < /dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r06/mod8k/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.3f" % (float(r["AverageNs"]) / 1e6)).rjust(10), "ms avg", ("%.1f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10), "ms total", r["Percentage"])
f = glob.glob("$R/gpurun_out/r06/mod8k/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("jxlhip::", "").replace("void ", "")[:40], r.get("Queue_Id", "")) for r in rows)
# last 420 ms of the run, kernels >= 3 ms, plus a summary line of the squeeze launches per 20 ms bin
end = ks[-1][1]
for s, e, n, q in ks:
    if s < end - 420e6 or e - s < 3e6: continue
    print(f"{(s - end) / 1e6 + 420:8.1f} -> {(e - end) / 1e6 + 420:8.1f} ({(e - s) / 1e6:7.2f}) q{q} {n}")
sq = [(s, e) for s, e, n, q in ks if "Squeeze" in n and s >= end - 420e6]
print("squeeze launches in the window:", len(sq), "busy span ms:", round(sum(e - s for s, e in sq) / 1e6, 1))
PY
grep "^{" $R/gpurun_out/r06/mod8k/log.txt
find $R/gpurun_out/r06/mod8k -name "*trace.csv" -size +5M -delete
