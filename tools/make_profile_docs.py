#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a round into the tracked summaries under profiles/.
Usage: make_profile_docs.py <tag> <stats_dir> <fetch_dir> <write_dir> <bench_log> [frames_in_pmc_launch]
  stats_dir : rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline
  fetch_dir / write_dir : rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --batch N --steps 1 --warmup 0 --no-pipeline --no-cpu-baseline
Writes the two markdown tables profiles/<tag>_stats_table.md / <tag>_pmc_table.md (pasted into the round's profile notes) and
profiles/pmc_traffic.json (read by bench.py)."""
import csv, glob, json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, fetch_dir, write_dir, log = sys.argv[1:6]
frames = int(sys.argv[6]) if len(sys.argv) > 6 else 32
write_json = not (len(sys.argv) > 7 and sys.argv[7] == "no-json")     # (the tables of the 8K workloads: pmc_traffic.json stays the 4K one bench.py reads)

def pmc(d):
    out = {}
    txt = subprocess.check_output([sys.executable, os.path.join(R, "tools", "pmc_sum.py"), d]).decode().strip().split("\n")
    for l in txt[1:]:
        k, c, v = l.rsplit(",", 2)
        out[k] = float(v)
    return out

F, W = pmc(fetch_dir), pmc(write_dir)
per, rows = {}, []
for k in sorted(set(F) | set(W), key=lambda k: -(2 * F.get(k, 0) + W.get(k, 0))):
    if k.startswith("__amd"):
        continue
    t = (2 * F.get(k, 0) + W.get(k, 0)) * 1024 / frames
    kk = k.replace("<true>", "")
    per[kk] = int(t)
    rows.append((kk, int(F.get(k, 0)), int(W.get(k, 0)), t / 1e6))
for k in list(per):          # template instantiations also under their plain name (bench.py looks kernels up by it)
    if "<" in k:
        base = k.split("<")[0]
        per[base] = max(per.get(base, 0), per[k])
if write_json:
  json.dump({"unit": "bytes per frame per launch (3840x2160 VarDCT d1, u8 RGB out)", "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / frames; separate --pmc passes",
           "frames_in_profiled_launch": frames, "per_kernel": per}, open(os.path.join(R, "profiles", "pmc_traffic.json"), "w"), indent=1)
md = "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB / frame |\n|---|---|---|---|\n" + "".join("| %s | %d | %d | %.2f |\n" % r for r in rows)
open(os.path.join(R, "profiles", tag + "_pmc_table.md"), "w").write(md)
d = json.loads([l for l in open(log) if '"metric"' in l][-1])
srows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0])))
md = "bench line of the profiled run: %.0f Mpixel/s, %.2f ms per step, stage_ms %s\n\n| kernel | calls | total ms | average ms | %% | min ms | max ms |\n|---|---|---|---|---|---|---|\n" % (d["value"], d["ms_per_step"], json.dumps(d["stage_ms"]))
for r in srows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("jxlhip::", "")
    md += "| %s | %s | %.1f | %.3f | %s | %.3f | %.3f |\n" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"], float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
open(os.path.join(R, "profiles", tag + "_stats_table.md"), "w").write(md)
print(md)
