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
