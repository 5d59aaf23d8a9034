#!/usr/bin/env python3
"""Per-kernel call count / average / total duration from a rocprofv3 kernel_trace.csv (for passes that ran without --stats, e.g. the PMC passes: kernels run one at a time there,
so these are stand-alone durations).  usage: trace_stats.py <dir or csv> [title]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("jxlhip::", "").replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(n, [0, 0.0, 1e30, 0.0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
if len(sys.argv) > 2: print(sys.argv[2] + "\n")
print("| kernel | calls | total ms | average ms | min ms | max ms |\n|---|---|---|---|---|---|")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[1] < 0.05: continue
    print(f"| {n} | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.3f} | {a[2]:.3f} | {a[3]:.3f} |")
