"""discorpy_amd.stack with the HIP kernels doing the per-shard work (tests/test_stack_distributed.py runs the same functions on CPU
with the oracle injected): one process without a process group, and two ranks sharing the test box's one GPU over gloo --
float32 and uint16 shards (the latter cross the collective as byte views), plain and pipelined, every rank's result against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, noise, typed_image

pytestmark = pytest.mark.gpu

ARGS = (250.3, 140.8, [1.0, 3.0e-5, -4.0e-8, 1e-11, -2e-14])
SHAPE = (300, 517)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _volumes(depth):
    return noise(91, (depth,) + SHAPE), typed_image("uint16", (depth, SHAPE[0], SHAPE[1] + 3), 92)


def test_one_process_no_group_equals_the_oracle(hip, orc):
    torch = pytest.importorskip("torch")
    from discorpy_amd import stack
    depth = 9
    vol, vol16 = _volumes(depth)
    got = stack.unwarp_stack_sharded(torch.from_numpy(vol).cuda(), depth, *ARGS, 7, 200)
    assert "stack" in hip.last_kernel()
    assert np.array_equal(got.cpu().numpy(), orc.unwarp_stack_rows(vol, *ARGS, 7, 200, coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))
    got16 = stack.unwarp_stack_sharded(torch.from_numpy(vol16).cuda(), depth, *ARGS, 7, 200)
    assert got16.dtype == torch.uint16
    assert np.array_equal(got16.cpu().numpy(), orc.unwarp_chunk_slices_backward(vol16, *ARGS, 7, 206, poly=orc.POLY_KERNEL))


def _worker(rank, world, port, depth, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from discorpy_amd import stack
    torch.cuda.set_device(0)                       # both ranks on the box's one GPU (RCCL refuses that; gloo carries the blocks)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol, vol16 = _volumes(depth)
        d0, d1 = stack.shard_bounds(depth, world, rank)
        res = {}
        for name, v in (("f32", vol), ("u16", vol16)):
            local = torch.from_numpy(v[d0:d1].copy()).cuda()
            res[name] = stack.unwarp_stack_sharded(local, depth, *ARGS, 7, 200).cpu()
            res[name + "_piped"] = stack.unwarp_stack_sharded(local, depth, *ARGS, 7, 200, pipeline=2).cpu()
            res[name + "_part"] = stack.unwarp_stack_sharded(local, depth, *ARGS, 7, 200, gather=False).cpu()
        np.savez(os.path.join(result_dir, "rank%d.npz" % rank), d0=d0, d1=d1, **{k: t.numpy() for k, t in res.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth", [8, 7])       # even and ragged shards
def test_two_ranks_on_one_gpu_over_gloo(tmp_path, hip, orc, depth):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, depth, str(tmp_path)), nprocs=world, join=True)
    vol, vol16 = _volumes(depth)
    want = {"f32": orc.unwarp_stack_rows(vol, *ARGS, 7, 200, coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP),
            "u16": orc.unwarp_chunk_slices_backward(vol16, *ARGS, 7, 206, poly=orc.POLY_KERNEL)}
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        d0, d1 = int(z["d0"]), int(z["d1"])
        for name in ("f32", "u16"):
            assert z[name].dtype == want[name].dtype
            assert np.array_equal(z[name], want[name]), (rank, name)
            assert np.array_equal(z[name + "_piped"], want[name]), (rank, name, "pipelined")
            assert np.array_equal(z[name + "_part"], want[name][d0:d1]), (rank, name, "local block")
