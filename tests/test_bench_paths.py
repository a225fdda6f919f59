"""CPU suite: the multi-rank code path of bench.py -- stack_scaling(), the function the driver's N = 1, 2, 4, 8 runs go
through for BASELINE config 4 -- with world_size 2 on gloo.  The HIP kernel cannot run here, so the device hooks
(make_block / launch / sync / fill) are CPU stand-ins and the per-shard work is the oracle; what is under test is the
sharding, the barrier + max-over-ranks timing protocol, the all-gather into the depth-outer result and the result dict."""
import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


CFG = dict(name="cfg4_small", shape=(6, 40, 56), seed=5, xcenter=27.5, ycenter=19.25, list_fact=[1.01, -2e-3, 3e-5])
NROWS = 40


class CpuBlock:
    def __init__(self, shape):
        import torch
        self.shape = tuple(shape)
        self.tensor = torch.zeros(self.shape, dtype=torch.float32)
        self.ptr = self.tensor.data_ptr()

    def free(self):
        pass


def _worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    from discorpy_amd import stack as st
    from oracle import oracle as orc
    orc.build()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, H, W = CFG["shape"]
        whole = np.random.default_rng(99).random((D, H, W), dtype=np.float32)
        d0, d1 = st.shard_bounds(D, world, rank)

        def fill(block, seed):
            block.tensor.copy_(torch.from_numpy(whole[d0:d1]))
            return whole[d0:d1]

        def launch(vol, out, dl):
            res = orc.unwarp_stack_rows(vol.tensor.numpy(), CFG["xcenter"], CFG["ycenter"], CFG["list_fact"], 0, NROWS,
                                        coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
            out.tensor.copy_(torch.from_numpy(res))

        want = orc.unwarp_stack_rows(whole, CFG["xcenter"], CFG["ycenter"], CFG["list_fact"], 0, NROWS, coord_round_f32=True,
                                     poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)

        def verify(chunk, out, full, d0_, dl):
            ok = np.array_equal(out.tensor.numpy(), want[d0_:d0_ + dl])
            return ok and full is not None and np.array_equal(full.tensor.numpy(), want)      # every rank holds the whole block

        res = bench.stack_scaling(CFG, world, rank, dist, steps=2, warmup=1, nrows=NROWS, make_block=CpuBlock, launch=launch,
                                  sync=lambda: None, fill=fill, barrier_device="cpu", verify=verify)
        json.dump(res, open(os.path.join(result_dir, "rank%d.json" % rank), "w"))
    finally:
        dist.destroy_process_group()


def test_stack_scaling_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]
    for r in res:
        assert r["verified_vs_oracle"] is True
        assert r["depth_per_gpu"] == 3 and r["compute_plus_allgather"] is not None
        assert r["compute_plus_allgather"]["gathered_bytes_received_per_gpu"] == 3 * NROWS * 56 * 4
        assert r["compute_only"]["ms_per_step"] > 0 and r["compute_plus_allgather"]["ms_per_step"] > 0
        # the pipelined exchange (depth sub-blocks, asynchronous all-gathers under the next kernel) fills the same result
        pp_ = r["compute_plus_allgather_pipelined"]
        assert pp_["verified_vs_oracle"] is True and pp_["depth_sub_blocks"] == 3 and pp_["ms_per_step"] > 0
    # max over ranks: both ranks report the same times
    assert res[0]["compute_only"]["ms_per_step"] == pytest.approx(res[1]["compute_only"]["ms_per_step"])


def test_stack_scaling_one_rank_has_no_collective():
    import bench
    from oracle import oracle as orc
    orc.build()
    D, H, W = CFG["shape"]
    whole = np.random.default_rng(98).random((D, H, W), dtype=np.float32)

    def fill(block, seed):
        import torch
        block.tensor.copy_(torch.from_numpy(whole))
        return whole

    def launch(vol, out, dl):
        import torch
        out.tensor.copy_(torch.from_numpy(orc.unwarp_stack_rows(vol.tensor.numpy(), CFG["xcenter"], CFG["ycenter"], CFG["list_fact"], 0, NROWS,
                                                                coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)))
    res = bench.stack_scaling(CFG, 1, 0, None, steps=1, warmup=0, nrows=NROWS, make_block=CpuBlock, launch=launch, sync=lambda: None,
                              fill=fill, barrier_device="cpu", verify=lambda chunk, out, full, d0, dl: full is None and dl == D)
    assert res["compute_plus_allgather"] is None and res["verified_vs_oracle"] is True and res["depth_per_gpu"] == D


def test_bench_argument_defaults_finish_in_minutes():
    import bench
    a = bench.parse([])
    assert (a.gpus, a.workload, a.batch, a.blend, a.order) == (1, "frame", 24, "f64lerp", 1) and a.steps * a.batch <= 5000
