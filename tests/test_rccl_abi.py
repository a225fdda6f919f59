"""The torch-free exchange of the depth-sharded stack: dcp_rccl_* / dcp_unwarp_stack_rows_rccl_f32 (dlopen'ed librccl, ncclAllGather
in place behind the kernel; pipelined: grouped ncclBroadcasts of depth sub-blocks on a side stream).  The test boxes have ONE GPU
and RCCL refuses two ranks on one device, so the hardware test runs at world size 1 (where the all-gather is the identity and the
pipelined path still cuts the shard into sub-blocks); world > 1 is covered by construction -- depth-outer layout, in-place
send = recv + rank * count -- and by bench.py's child processes on the first multi-GPU node.  Reference: the loops over depth of
postprocessing.py:226-228, 310-312 carry no state."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, noise

from discorpy_amd import _ffi as F


def test_rccl_entry_points_validate_without_a_gpu():
    L = F.lib()
    assert L.dcp_rccl_available() in (0, 1)
    one = (C.c_double * 1)(1.0)
    buf = np.zeros((2, 4, 4), np.float32)
    assert L.dcp_unwarp_stack_rows_rccl_f32(buf.ctypes.data, buf.ctypes.data, 2, 4, 4, 16, 4, 1.0, 1.0, one, 1, 0.0, 1, 1, F.BLEND_F64LERP,
                                            None, 1, None) == F.ERR_INVALID_ARG
    comm = C.c_void_p()
    small = (C.c_char * 16)()
    assert L.dcp_rccl_unique_id(small, 16) == F.ERR_INVALID_ARG
    assert L.dcp_rccl_comm_create(C.byref(comm), 2, 5, small, -1) == F.ERR_INVALID_ARG
    assert L.dcp_rccl_comm_destroy(None) == F.OK


@pytest.mark.gpu
def test_rccl_allgather_world_of_one_equals_the_plain_stack_call(hip, orc):
    L = hip.lib()
    assert L.dcp_rccl_available() == 1, hip.last_error()
    D, H, W, nrows, row0 = 7, 300, 520, 64, 100.0
    vol = noise(77, (D, H, W))
    xc, yc, fact = 250.3, 160.7, [1.0, -2e-5, 3e-8]
    fa, nf = hip.fact_array(fact)
    dvol = hip.DeviceBuffer(vol.nbytes).upload(vol)
    dout = hip.DeviceBuffer(D * nrows * W * 4)
    idbuf = (C.c_char * 128)()
    hip.check(L.dcp_rccl_unique_id(idbuf, 128))
    comm = C.c_void_p()
    hip.check(L.dcp_rccl_comm_create(C.byref(comm), 1, 0, idbuf, -1))
    want = orc.unwarp_stack_rows(vol, xc, yc, fact, row0, nrows, coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
    try:
        for pipeline in (1, 3, 50):
            dout.upload(np.zeros((D, nrows, W), np.float32))
            hip.check(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, D, H, W, H * W, W, xc, yc, fa, nf, row0, nrows, 1, hip.BLEND_F64LERP,
                                                       comm, pipeline, None))
            hip.check(L.dcp_stream_synchronize(-1, None))
            assert np.array_equal(dout.download((D, nrows, W), np.float32), want), pipeline
    finally:
        hip.check(L.dcp_rccl_comm_destroy(comm))


def _run_child(args, env=None, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stderr or r.stdout)[-2000:]
    return json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_children_of_the_exchange_without_torch(hip, tmp_path):
    # the native RCCL rank process bench.py starts per GPU at N > 1 (here a world of one), checked against the oracle inside
    j = _run_child(["--native-child", "rccl", "--child-world", "1", "--child-rank", "0", "--idfile", str(tmp_path / "id"), "--depth", "8",
                    "--rows", "2560", "--steps", "2"])
    assert j["allgather_verified"] is True and j["allgather_pipelined_verified"] is True and j["allgather_ms"] > 0
    # the peer-copy process: two slots on the one GPU (DCP_BENCH_DEVICE hook)
    j = _run_child(["--native-child", "peer", "--child-world", "2", "--depth", "8", "--rows", "2560", "--steps", "2"], env={"DCP_BENCH_DEVICE": "0"})
    assert j["verified"] is True and j["devices"] == [0, 0] and j["peer_copies_ms"] > 0
