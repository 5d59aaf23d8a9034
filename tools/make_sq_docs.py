#!/usr/bin/env python3
"""profiles/sq_valu.json (read by bench.py's valu_issue) from an SQ counter pass of a round:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d <dir> -- python tools/experiments/one_batch_decode.py 4k <frames> <decodes>
Usage: make_sq_docs.py <dir> <frames> <decodes> [round label]"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, frames, decodes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else "round 5"
txt = subprocess.check_output([sys.executable, os.path.join(R, "tools", "pmc_sum.py"), d]).decode().strip().split("\n")
cols = txt[0].split(",")
per = {}
for l in txt[1:]:
    parts = l.rsplit(",", len(cols) - 1)
    k, calls, vals = parts[0], int(parts[1]), dict(zip(cols[2:], map(float, parts[2:])))
    if k.startswith("__amd") or "at::native" in k:
        continue
    per[k.replace("<true>", "") if k.endswith("Kernel<true>") else k] = {
        "valu_wave_instr_per_frame": vals.get("SQ_INSTS_VALU", 0.0) / frames / max(1, calls), "calls": calls, "waves_per_frame": vals.get("SQ_WAVES", 0.0) / frames / max(1, calls)}
json.dump({"what": f"SQ_INSTS_VALU (wavefront-level VALU instructions) per 3840x2160 frame and kernel: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over {decodes} plain decodes of {frames} frames "
                   f"(tools/scripts/profile_round5.sh, {label}); kernels with calls == {decodes} ran in every decode (bench.py sums those)", "per_kernel": per},
          open(os.path.join(R, "profiles", "sq_valu.json"), "w"), indent=1)
print(json.dumps({k: round(v["valu_wave_instr_per_frame"]) for k, v in per.items() if v["calls"] == decodes}, indent=1))
