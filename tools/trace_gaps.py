#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per kernel name start/end relative to the first HF kernel of each step.
Usage: trace_gaps.py <kernel_trace.csv> [first_step] [n_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("jxlhip::", "").replace("void ", "")[:28], r.get("Queue_Id", "")) for r in rows]
ks.sort()
hf = [i for i, k in enumerate(ks) if k[2].startswith("HfDecodeSimt")]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(hf) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0 = ks[hf[first]][0]
t1 = ks[hf[first + n]][0] if first + n < len(hf) else ks[-1][1]
print(f"steps {first}..{first+n-1}: {(t1 - t0) / 1e6 / n:.2f} ms per step")
for s, e, name, q in ks:
    if s < t0 - 20e6 or s > t1: continue
    if (e - s) < 0.3e6 and not name.startswith(("Hf", "Lf")): continue
    print(f"{(s - t0) / 1e6:9.2f} -> {(e - t0) / 1e6:9.2f}  ({(e - s) / 1e6:7.2f} ms)  q{q}  {name}")
