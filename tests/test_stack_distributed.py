"""CPU suite, part 3: the N > 1 path -- depth sharding + all-gather -- with world_size 2 on gloo.

The HIP kernel cannot run here, so `compute=` injects the CPU oracle for the per-shard work; what
is under test is discorpy_amd.stack (shard bounds, ragged shards, the collective, ordering)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, noise, typed_image

from discorpy_amd import stack


def test_shard_bounds_cover_the_depth_exactly():
    for depth in (0, 1, 7, 8, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [stack.shard_bounds(depth, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == depth
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        stack.shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, depth, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol = noise(77, (depth, 40, 56))
        d0, d1 = stack.shard_bounds(depth, world, rank)
        local = torch.from_numpy(vol[d0:d1].copy())

        def cpu_rows(lv, xc, yc, fact, row_start, nrows, round32, blend):
            return orc.unwarp_stack_rows(lv.numpy(), xc, yc, fact, row_start, nrows, coord_round_f32=round32,
                                         poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)

        args = (27.5, 19.25, [1.01, -2e-3, 3e-5])
        full = stack.unwarp_stack_sharded(local, depth, *args, 10, 6, coord_round_f32=True, compute=cpu_rows)
        part = stack.unwarp_stack_sharded(local, depth, *args, 10, 6, coord_round_f32=True, gather=False,
                                          compute=cpu_rows)
        sl = stack.unwarp_stack_sharded(local, depth, *args, 21, 1, coord_round_f32=False, compute=cpu_rows)
        # the all-gather pipelined against the per-shard work in depth sub-blocks (falls back for ragged shards)
        piped = stack.unwarp_stack_sharded(local, depth, *args, 10, 6, coord_round_f32=True, compute=cpu_rows, pipeline=2)
        piped9 = stack.unwarp_stack_sharded(local, depth, *args, 10, 6, coord_round_f32=True, compute=cpu_rows, pipeline=9)
        # the sharding without a collective: every rank holds the stack and owns output rows of every projection
        rows, (r0, r1) = stack.unwarp_stack_row_sharded(torch.from_numpy(vol), *args, 3, 29, coord_round_f32=True, compute=cpu_rows)
        # a uint16 stack (what detectors deliver): torch's uint16 is not an element type of the collectives -- it travels as int16 views
        vol16 = typed_image("uint16", (depth, 40, 56), 78)
        local16 = torch.from_numpy(vol16[d0:d1].copy())

        def cpu_rows16(lv, xc, yc, fact, row_start, nrows, round32, blend):
            return torch.from_numpy(orc.unwarp_chunk_slices_backward(lv.numpy(), xc, yc, fact, int(row_start), int(row_start) + nrows - 1, poly=orc.POLY_KERNEL))

        full16 = stack.unwarp_stack_sharded(local16, depth, *args, 10, 6, coord_round_f32=True, compute=cpu_rows16)
        piped16 = stack.unwarp_stack_sharded(local16, depth, *args, 10, 6, coord_round_f32=True, compute=cpu_rows16, pipeline=2)
        assert full16.dtype == torch.uint16 and piped16.dtype == torch.uint16
        np.savez(os.path.join(result_dir, "rank%d.npz" % rank), full=full.numpy(), part=part.numpy(), sl=sl.numpy(),
                 piped=piped.numpy(), piped9=piped9.numpy(), d0=d0, d1=d1, rows=np.asarray(rows), r0=r0, r1=r1,
                 full16=full16.numpy(), piped16=piped16.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth", [6, 7])       # even and ragged shards
def test_two_rank_gloo_all_gather_reassembles_the_stack(tmp_path, orc, depth):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, depth, str(tmp_path)), nprocs=world, join=True)
    vol = noise(77, (depth, 40, 56))
    args = (27.5, 19.25, [1.01, -2e-3, 3e-5])
    want = orc.unwarp_stack_rows(vol, *args, 10, 6, coord_round_f32=True, poly=orc.POLY_KERNEL,
                                 blend=orc.BLEND_F64LERP)
    want_sl = orc.unwarp_stack_rows(vol, *args, 21, 1, coord_round_f32=False, poly=orc.POLY_KERNEL,
                                    blend=orc.BLEND_F64LERP)
    want_rows = orc.unwarp_stack_rows(vol, *args, 3, 29, coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
    want16 = orc.unwarp_chunk_slices_backward(typed_image("uint16", (depth, 40, 56), 78), *args, 10, 15, poly=orc.POLY_KERNEL)
    covered = []
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        r0, r1 = int(z["r0"]), int(z["r1"])
        assert z["rows"].shape == (depth, r1 - r0, 56) and np.array_equal(z["rows"], want_rows[:, r0:r1])      # row-sharded: no collective
        covered += list(range(r0, r1))
    assert covered == list(range(29))
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert np.array_equal(z["full"], want)                       # every rank holds the whole block
        assert np.array_equal(z["part"], want[int(z["d0"]):int(z["d1"])])
        assert np.array_equal(z["sl"], want_sl)
        assert np.array_equal(z["piped"], want) and np.array_equal(z["piped9"], want)
        assert z["full16"].dtype == np.uint16 and np.array_equal(z["full16"], want16) and np.array_equal(z["piped16"], want16)
