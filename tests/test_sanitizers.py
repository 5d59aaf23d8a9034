"""The host half of the decoder under AddressSanitizer + UndefinedBehaviorSanitizer (not gpu): every host source of the library is rebuilt with g++ -fsanitize=address,undefined
(jpegxl-rs_amd/Makefile target `asan`) and the mutation fuzzer tests/fuzz/fuzz_host.cc replays the fixtures, the goldens and fresh synthesised streams — as they are, then with bit
flips, truncations, splices and length-field edits — through everything that parses untrusted bytes on the host: container and box walk, image / frame headers, TOC, MA trees,
histograms, quantisation tables, coefficient orders, patches / splines / noise syntax, ICC stream, the jbrd box and JPEG marker rebuild, and the JxlDecoder state machine up to the
point where it needs a device.  The reference runs its suite under ASan / TSan in CI (/root/reference/.github/workflows/ci.yml:71-106); the code it sanitises there
is libjxl's — this is ours.  A short budget here; tools/scripts/fuzz_host_long.sh runs >= 10^5 trials."""
import json
import os
import subprocess

import pytest

from conftest import ROOT, FIXTURES, GOLDEN
import synth_lib as S

PKG = os.path.join(ROOT, "jpegxl-rs_amd")


@pytest.fixture(scope="module")
def fuzzer(built):
    subprocess.check_call(["make", "-s", "-C", PKG, "-j8", "asan"])
    exe = os.path.join(PKG, "build_asan", "fuzz_host")
    assert os.path.exists(exe)
    return exe


def make_corpus(path):
    import numpy as np
    os.makedirs(path, exist_ok=True)
    n = 0
    for d in (FIXTURES, GOLDEN):
        for name in sorted(os.listdir(d)):
            if name.endswith(".jxl"):
                with open(os.path.join(d, name), "rb") as f, open(os.path.join(path, f"{n:03d}_{name}"), "wb") as g:
                    g.write(f.read())
                n += 1
    img = S.synthetic_image(77, 200, 136)
    extra = [S.encode_vardct(img, seed=1, strategy_mix=2, epf_iters=1, gab=1), S.encode_vardct(img, seed=2, strategy_mix=1, num_passes=2, permute_toc=3),
             S.encode_ycbcr(img, subsampling="420", seed=3), S.encode_modular(img.astype(np.int32), 8, True, 1),
             S.encode_modular_free(seed=5, w=96, h=80, tree_flags=3, local_trees=1, lz77=True)]
    for k, data in enumerate(extra):
        with open(os.path.join(path, f"{n + k:03d}_synth.jxl"), "wb") as g:
            g.write(data)
    return n + len(extra)


def test_host_parser_under_asan_ubsan(fuzzer, tmp_path):
    corpus = str(tmp_path / "corpus")
    count = make_corpus(corpus)
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=1:detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    procs = [subprocess.Popen([fuzzer, corpus, "1500", "40", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for seed in (1, 2, 3, 4)]
    total = accepted = 0
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-4000:]
        d = json.loads(out.strip().splitlines()[-1])
        assert d["corpus"] == count
        total += d["trials"]; accepted += d["accepted"]
    assert total >= 2000 and accepted >= 4 * count          # (every unmutated corpus file is accepted in each process)
