#!/usr/bin/env python3
"""profiles/<tag>_latency_table.md from the latency passes of tools/scripts/profile_round6.sh: per kernel of a ONE-frame decode (one stream per wavefront) its average duration
(rocprofv3 --kernel-trace --stats) and the wavefront-level VALU / SALU instructions and wavefronts per launch (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES).
Usage: make_latency_docs.py <tag> <gpurun_out dir of the round>"""
import csv, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]

def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("jxlhip::", "")
    return n

def stats(sub):
    out = {}
    p = os.path.join(d, sub, "p_kernel_stats.csv")
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        out[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    return out

def counters(sub):
    agg = {}
    p = os.path.join(d, sub, "p_counter_collection.csv")
    if not os.path.exists(p):
        return agg
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, {"disp": set()})
        a["disp"].add(r["Dispatch_Id"])
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg

lines = []
for sub, title in (("lat0", "one 3840x2160 frame, gradient LF tree (the headline's frames)"), ("lat1", "one 3840x2160 frame, weighted-predictor LF tree (cjxl's default shape)")):
    st, ct = stats(sub + "_stats"), counters(sub + "_SQ")
    lines += [f"### {title}: LF stride 64, hf_lanes_per_wave 1 (`one_batch_decode.py 4k 1 4`)", "", "| kernel | calls | avg ms | VALU wave-instr per launch | SALU wave-instr per launch | wavefronts per launch |", "|---|---|---|---|---|---|"]
    for k, (calls, ms) in sorted(st.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        if k.startswith("__amd") or "at::native" in k:
            continue
        c = ct.get(k, {})
        n = max(1, len(c.get("disp", [])))
        lines.append(f"| {k} | {calls} | {ms:.3f} | {c.get('SQ_INSTS_VALU', 0) / n:,.0f} | {c.get('SQ_INSTS_SALU', 0) / n:,.0f} | {c.get('SQ_WAVES', 0) / n:,.0f} |")
    lines.append("")
st = stats("benchjxl_stats")
if st:
    lines += ["### samples/bench.jxl (lossless Modular 2122x1433 RGBA, the reference's criterion input) through decode_with, three decodes", "", "| kernel | calls | avg ms |", "|---|---|---|"]
    for k, (calls, ms) in sorted(st.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        if not (k.startswith("__amd") or "at::native" in k):
            lines.append(f"| {k} | {calls} | {ms:.3f} |")
    lines.append("")
open(os.path.join(R, "profiles", f"{tag}_latency_table.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
