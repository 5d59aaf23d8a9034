"""prints the numbers of a bench.py JSON line that the notes quote (headline, legs, latency, API)  usage: summarise_bench_line.py file.json [...]"""
import json, sys
for fn in sys.argv[1:]:
    d = json.loads(open(fn).read().strip().splitlines()[-1])
    c = d["config"]
    print(fn, d["value"], "Mpixel/s", d["ms_per_step"], "ms/step, steady", d.get("steady_state_ms_per_step"), d.get("stage_ms"))
    for k in c:
        if k.startswith("workload_"):
            v = c[k]
            print("  ", k, {kk: v[kk] for kk in v if kk in ("value", "ms_per_step", "steady_state_ms_per_step", "single_image_ms", "jobs_in_flight", "error", "stage_ms")})
    s = d.get("single_frame_ms", {})
    print("   single", {kk: vv for kk, vv in s.items() if kk.endswith("_ms") or kk == "value"})
    print("   one_pass", {kk: vv for kk, vv in d.get("one_pass_128", {}).items() if kk != "what"})
    print("   pcie_inclusive", d.get("pcie_inclusive", {}).get("mpixel_per_s"), " streaming_host_out", {k: d.get("streaming_host_out", {}).get(k) for k in ("value", "fraction_of_pcie_ceiling")})
    print("   api", {kk: (vv.get("mpixel_per_s"), vv.get("median_latency_ms")) for kk, vv in d.get("api_concurrent", {}).items() if kk.startswith("threads")})
    print("   roofline", d.get("roofline")); print("   hbm stage", d.get("roofline_hbm_stage")); print("   cpu", d.get("cpu_baseline"))
    for k in ("extras_error",):
        if k in d: print("  ", k, d[k])
