"""The gather of decoded pixels behind the C ABI (include/jxl_hip.h JxlHipComm* / JxlHipGatherFrames*; csrc/gather.cc): the one exchange step of the multi-GPU path
 without PyTorch on the caller's side.  not gpu: the symbols are exported and declared.  -m gpu: a world of one rank (device copy into the job buffer,
equal and ragged shards, the checksum all-reduce) and the RCCL library loads and hands out a communicator id.  A world of several ranks needs several GPUs: the
driver's scaling run is the first execution of that path (the same point-to-point pattern runs through torch.distributed in tests/test_multiprocess_gloo.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def jx(built):
    import jpegxl_rs_amd as jx
    return jx


def test_gather_symbols_exported_and_declared(jx):
    L = jx.libjxl()
    header = open(os.path.join(ROOT, "include", "jxl_hip.h")).read()
    for name in ("JxlHipCommGetUniqueId", "JxlHipCommCreate", "JxlHipCommDestroy", "JxlHipGatherFrames", "JxlHipGatherFramesRagged", "JxlHipAllReduceSumI64"):
        assert hasattr(L, name) and name in header, name
    assert "JXL_HIP_COMM_ID_BYTES 128" in header


def test_fake_rccl_double_exports_what_gather_cc_binds(built):
    """the shared-memory double of librccl (tests/fake_rccl, built by tools/Makefile) carries every entry point csrc/gather.cc looks up"""
    fake = os.path.join(ROOT, "tools", "_build", "libfake_rccl.so")
    assert os.path.exists(fake), "make -C tools"
    L = C.CDLL(fake)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclAllReduce", "ncclGetErrorString"):
        assert hasattr(L, name), name
    uid = (C.c_uint8 * 128)()
    assert L.ncclGetUniqueId(uid) == 0 and bytes(uid).startswith(b"fake_rccl_")


@pytest.mark.gpu
def test_gather_world_of_one_and_rccl_loads(jx):
    import torch
    assert torch.cuda.is_available()
    L = jx.libjxl()
    uid = (C.c_uint8 * 128)()
    assert L.JxlHipCommGetUniqueId(uid) == 0, jx.last_error()          # librccl.so loads and answers
    assert any(uid)
    comm = L.JxlHipCommCreate(0, 0, 1, uid)
    assert comm
    assert not L.JxlHipCommCreate(0, 1, 1, uid)                         # rank outside the world
    frames, fb = 5, 640 * 480 * 3
    send = torch.randint(0, 255, (frames, fb), dtype=torch.uint8, device="cuda")
    recv = torch.zeros((1, frames, fb), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.JxlHipGatherFrames(comm, send.data_ptr(), fb, frames, recv.data_ptr(), 0, 2, st) == 0, jx.last_error()
    torch.cuda.synchronize()
    assert torch.equal(recv[0], send)
    recv.zero_()
    per = (C.c_int * 1)(3)
    assert L.JxlHipGatherFramesRagged(comm, send.data_ptr(), fb, per, recv.data_ptr(), 0, 32, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(recv[0][:3], send[:3]) and int(recv[0][3:].sum()) == 0
    v = torch.tensor([41, 1], dtype=torch.int64, device="cuda")
    assert L.JxlHipAllReduceSumI64(comm, v.data_ptr(), 2, st) == 0
    torch.cuda.synchronize()
    assert v.tolist() == [41, 1]
    assert L.JxlHipGatherFrames(comm, send.data_ptr(), fb, frames, None, 0, 2, st) == 1 and "missing buffer" in jx.last_error()
    L.JxlHipCommDestroy(comm)


# ---- world > 1 on ONE GPU: csrc/gather.cc over a test double of librccl (tests/fake_rccl: mailboxes in shared memory, device data staged through the host) ------------------
_WORKER = r'''
import ctypes as C, os, sys, json
rank, world, idfile, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
sys.path.insert(0, os.environ["JXL_REPO_ROOT"])
import torch
import jpegxl_rs_amd as jx
L = jx.libjxl()
uid = (C.c_uint8 * 128)()
if rank == 0:
    assert L.JxlHipCommGetUniqueId(uid) == 0, jx.last_error()
    open(idfile + ".tmp", "wb").write(bytes(uid)); os.replace(idfile + ".tmp", idfile)
else:
    import time
    for _ in range(600):
        if os.path.exists(idfile): break
        time.sleep(0.05)
    uid = (C.c_uint8 * 128).from_buffer_copy(open(idfile, "rb").read())
L.JxlHipCommCreate.restype = C.c_void_p
comm = L.JxlHipCommCreate(0, rank, world, uid)
assert comm, jx.last_error()
comm = C.c_void_p(comm)
fb = 1000 * 3 + 7                                  # (an odd frame size: nothing relies on alignment)
per = [5] * world if mode == "equal" else [3, 3, 2, 2][:world]
def frames_of(r):
    g = torch.Generator().manual_seed(1234 + r)
    return torch.randint(0, 256, (per[r], fb), dtype=torch.uint8, generator=g)
send = frames_of(rank).cuda()
total = sum(per)
recv = torch.zeros((total, fb), dtype=torch.uint8, device="cuda") if rank == 0 else None
st = torch.cuda.current_stream().cuda_stream
if mode == "equal":
    rc = L.JxlHipGatherFrames(comm, C.c_void_p(send.data_ptr()), C.c_size_t(fb), per[0], C.c_void_p(recv.data_ptr()) if rank == 0 else None, 0, 2, C.c_void_p(st))
else:
    arr = (C.c_int * world)(*per)
    rc = L.JxlHipGatherFramesRagged(comm, C.c_void_p(send.data_ptr()), C.c_size_t(fb), arr, C.c_void_p(recv.data_ptr()) if rank == 0 else None, 0, 2, C.c_void_p(st))
assert rc == 0, jx.last_error()
torch.cuda.synchronize()
ok = True
if rank == 0:
    want = torch.cat([frames_of(r) for r in range(world)])
    ok = bool(torch.equal(recv.cpu(), want))
v = torch.tensor([rank + 1, 10 * (rank + 1)], dtype=torch.int64, device="cuda")
assert L.JxlHipAllReduceSumI64(comm, C.c_void_p(v.data_ptr()), C.c_size_t(2), C.c_void_p(st)) == 0, jx.last_error()
torch.cuda.synchronize()
s = world * (world + 1) // 2
ok = ok and v.tolist() == [s, 10 * s]
L.JxlHipCommDestroy(comm)
print(json.dumps({"rank": rank, "ok": ok}))
'''


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "equal"), (4, "equal"), (4, "ragged")])
def test_gather_with_several_ranks_on_one_gpu(jx, tmp_path, world, mode):
    """JxlHipGatherFrames / ...Ragged / JxlHipAllReduceSumI64 with 2 and 4 ranks (ragged: 3, 3, 2, 2 frames), one process per rank, all on GPU 0, librccl replaced by
    tests/fake_rccl through JXL_HIP_RCCL_LIB: the id handed over by value (the double checks all 128 bytes), chunked groups of sends / receives (2 frames per chunk), every
    shard at its final position in rank 0's buffer, the checksum all-reduce."""
    import json, subprocess, sys
    fake = os.path.join(ROOT, "tools", "_build", "libfake_rccl.so")
    assert os.path.exists(fake), "tools/_build/libfake_rccl.so is missing (make -C tools)"
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, JXL_HIP_RCCL_LIB=fake, JXL_REPO_ROOT=ROOT)
    idfile = str(tmp_path / "comm_id")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), idfile, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {err[-2000:]}"
        assert json.loads(out.strip().splitlines()[-1]) == {"rank": r, "ok": True}
