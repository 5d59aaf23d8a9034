#!/bin/bash
# round 6: kernel stats of the JPEG-transcode-shaped leg alone (tools/experiments/gpu_r6_jpeg_leg.py, first job shape only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06/jpeg
cd /tmp && export TMPDIR=/tmp
JPEG_LEG_ONLY_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/jpeg -o j -- python $R/tools/experiments/gpu_r6_jpeg_leg.py > $R/gpurun_out/r06/jpeg/log.txt 2>&1 < /dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r06/jpeg/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.3f" % (float(r["AverageNs"]) / 1e6)).rjust(10), "ms avg", r["Percentage"])
PY
grep "^{" $R/gpurun_out/r06/jpeg/log.txt
find $R/gpurun_out/r06/jpeg -name "*trace.csv" -size +5M -delete
