#!/bin/bash
# VGPRs / spills / scratch / LDS of every kernel in lib/libjxl.so's gfx950 code object (llvm-readelf --notes on the extracted bundle)
set -e
cd "$(dirname "$0")/../.."
LIB=${1:-jpegxl-rs_amd/lib/libjxl.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB > $TMP/targets 2>/dev/null || true
python3 - "$LIB" "$TMP" <<'PY'
import sys, re, subprocess, os
lib, tmp = sys.argv[1], sys.argv[2]
data = open(lib, "rb").read()
# the fat binary holds ELF code objects: find every ELF header with the AMDGPU machine id (224)
out = []
pos = 0
k = 0
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0: break
    pos = i + 4
    if data[i + 18:i + 20] != b"\xe0\x00": continue
    # section header table end gives the size
    import struct
    shoff, = struct.unpack_from("<Q", data, i + 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    size = shoff + shentsize * shnum
    path = os.path.join(tmp, f"co{k}.elf"); k += 1
    open(path, "wb").write(data[i:i + size])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        out.append((m.group(1), int(m.group(4)), int(m.group(5)), int(m.group(2))))
flt = subprocess.run(["c++filt"] + [o[0] for o in out], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':90s} {'VGPRs':>6s} {'spilled':>8s} {'scratch B':>10s}")
for (n, v, sp, sc), d in sorted(zip(out, flt), key=lambda t: t[1]):
    d = re.sub(r"\(.*", "", d).replace("void jxlhip::", "")
    print(f"{d[:90]:90s} {v:6d} {sp:8d} {sc:10d}")
PY
rm -rf $TMP
