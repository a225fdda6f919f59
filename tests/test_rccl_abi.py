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
    assert L.dcp_rccl_comm_fixed_shards(None, 1) == F.ERR_INVALID_ARG
    info = (C.c_int64 * 10)()
    assert L.dcp_rccl_comm_info(None, info, 10, None, 0, None, 0) == F.ERR_INVALID_ARG


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
    def info_of():
        info, depths, path = (C.c_int64 * 10)(), (C.c_int64 * 1)(), C.create_string_buffer(1024)
        hip.check(L.dcp_rccl_comm_info(comm, info, 10, depths, 1, path, 1024))
        return list(info), int(depths[0]), path.value.decode()
    try:
        # what the communicator says about itself -- the REAL librccl here: one rank, rank 0, a version RCCL really has
        info, d_agreed, path = info_of()
        assert info[0] == 1 and info[1] == 0 and info[3] >= 20000 and info[4] == 1 and info[5] == 0 and info[2] == info[6]
        assert "librccl" in path and "fake" not in path and d_agreed == -1 and info[7] == 0 and info[8] == 0
        for pipeline in (1, 3, 50):
            dout.upload(np.zeros((D, nrows, W), np.float32))
            hip.check(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, D, H, W, H * W, W, xc, yc, fa, nf, row0, nrows, 1, hip.BLEND_F64LERP,
                                                       comm, pipeline, None))
            hip.check(L.dcp_stream_synchronize(-1, None))
            assert np.array_equal(dout.download((D, nrows, W), np.float32), want), pipeline
        info, d_agreed, _ = info_of()
        assert d_agreed == D and info[7] == 1 and info[8] == 3 and info[9] == 3
        # the caller vouches for unchanged shapes: no further agreements, the same result; other shapes are refused
        hip.check(L.dcp_rccl_comm_fixed_shards(comm, 1))
        dout.upload(np.zeros((D, nrows, W), np.float32))
        for _ in range(3):
            hip.check(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, D, H, W, H * W, W, xc, yc, fa, nf, row0, nrows, 1, hip.BLEND_F64LERP,
                                                       comm, 50, None))
        hip.check(L.dcp_stream_synchronize(-1, None))
        assert np.array_equal(dout.download((D, nrows, W), np.float32), want)
        info, _, _ = info_of()
        assert info[8] == 6 and info[9] == 3
        assert L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, D, H, W, H * W, W, xc, yc, fa, nf, row0, nrows - 1, 1, hip.BLEND_F64LERP,
                                                comm, 50, None) == hip.ERR_INVALID_ARG and "dcp_rccl_comm_fixed_shards" in hip.last_error()
        hip.check(L.dcp_rccl_comm_fixed_shards(comm, 0))
        hip.check(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, D, H, W, H * W, W, xc, yc, fa, nf, row0, nrows - 1, 1, hip.BLEND_F64LERP,
                                                   comm, 50, None))
        hip.check(L.dcp_stream_synchronize(-1, None))
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
    rc = j["rccl"]          # the communicator's own account (real RCCL at world 1)
    assert rc["nccl_comm_count"] == 1 and rc["nccl_comm_user_rank"] == 0 and rc["rccl_is_stand_in"] is False and rc["nccl_version_code"] >= 20000
    assert rc["shard_depths"] == [8] and "librccl" in rc["librccl"] and j["bytes_received"] == 0
    assert rc["agreements"] == 2 and rc["exchanges"] == 2 * (1 + 2)          # one agreement per variant, the timed calls repeat it
    # the peer-copy process: two slots on the one GPU (DCP_BENCH_DEVICE hook)
    j = _run_child(["--native-child", "peer", "--child-world", "2", "--depth", "8", "--rows", "2560", "--steps", "2"], env={"DCP_BENCH_DEVICE": "0"})
    assert j["verified"] is True and j["devices"] == [0, 0] and j["peer_copies_ms"] > 0
