"""The product's per-strategy geometry is packed into 64-bit immediates (branch-free lookups on the GPU,
jpegxl-rs_amd/csrc/jxl_dev.h); check it against the oracle's plain tables (oracle/vardct.h) for all 27 strategies."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_strategy_tables_match_the_oracle(tmp_path):
    exe = str(tmp_path / "host_tables")
    src = os.path.join(ROOT, "tests", "host_tables.cc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "jpegxl-rs_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           "-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout
    assert "0 mismatches" in out.stdout
