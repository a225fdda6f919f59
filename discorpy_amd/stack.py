"""Depth-sharded unwarp of a tomography stack across the GPUs of one node.

Every projection of a ``(depth, height, width)`` stack is independent
(reference ``discorpy/post/postprocessing.py:226-228, 310-312`` loop over ``depth`` with no carried
state), so rank ``g`` of ``G`` holds projections ``[d0, d1)``, runs the stack kernel on them, and
ONE exchange -- an all-gather along the depth axis (RCCL over xGMI with the ``nccl`` backend) --
reassembles the ``(depth, nrows, width)`` sinogram block on every rank.  Depth is the outermost
axis of the result, so each rank's contribution is one contiguous block.

One process per GPU (``torch.distributed``); the reference has no distributed code to mirror.
"""
import numpy as np


def shard_bounds(depth, world_size, rank):
    """Contiguous depth range [d0, d1) of `rank`; the first depth % world_size ranks get one more."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(depth), world_size)
    d0 = rank * base + min(rank, rem)
    return d0, d0 + base + (1 if rank < rem else 0)


def row_shard_bounds(nrows, world_size, rank):
    """Alternative with NO collective: rank owns output rows [r0, r1) of every projection."""
    return shard_bounds(nrows, world_size, rank)


def _hip_rows(local_vol, xcenter, ycenter, list_fact, row_start, nrows, coord_round_f32, blend):
    from .post import postprocessing as pp
    return pp._stack_rows(local_vol, xcenter, ycenter, list_fact, float(row_start), int(nrows),
                          bool(coord_round_f32), blend)


def _wire(t):
    """The tensor as the collective sees it.  RCCL / NCCL and gloo have no 16-bit integer element type, and torch's uint16 / uint32 /
    uint64 are not c10d element types either -- tomography detectors deliver uint16 -- so those stacks travel as the bytes they are
    (a uint8 view of the contiguous block: the last axis times the element size; the bits are the payload)."""
    import torch
    native = (torch.uint8, torch.int8, torch.int32, torch.int64, torch.float16, torch.bfloat16, torch.float32, torch.float64)
    return t if t.dtype in native else t.view(torch.uint8)


def unwarp_stack_sharded(local_vol, depth, xcenter, ycenter, list_fact, row_start, nrows, *,
                         coord_round_f32=True, gather=True, group=None, blend=None, compute=None, pipeline=1):
    """
    Rows ``row_start .. row_start+nrows-1`` of the corrected stack from a depth-sharded volume.

    Parameters
    ----------
    local_vol : torch.Tensor
        This rank's projections ``[d0, d1)`` (``shard_bounds(depth, world, rank)``), float32,
        shape ``(d1 - d0, height, width)``, on this rank's GPU.
    depth : int
        Depth of the whole stack.
    coord_round_f32 : bool
        True = ``unwarp_chunk_slices_backward`` semantics, False = ``unwarp_slice_backward``.
    gather : bool
        True: all-gather along depth, every rank returns ``(depth, nrows, width)``.
        False: return the local ``(d1 - d0, nrows, width)`` block only.
    compute : callable, optional
        Replaces the HIP kernel call (signature of ``_hip_rows``); used by the CPU ``gloo`` tests to
        exercise the sharding and the collective without a GPU.
    pipeline : int
        > 1 (even shards only): the local shard is cut into that many depth sub-blocks; the all-gather of
        sub-block ``s`` (asynchronous, on the collective's own stream) runs while the kernel of sub-block
        ``s + 1`` is computing, every rank's piece landing directly in its place of the result.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    d0, d1 = shard_bounds(depth, world, rank)
    if local_vol.shape[0] != d1 - d0:
        raise ValueError("rank %d holds %d projections, its shard of depth %d is [%d, %d)"
                         % (rank, local_vol.shape[0], depth, d0, d1))
    fn = _hip_rows if compute is None else compute
    counts = [shard_bounds(depth, world, r)[1] - shard_bounds(depth, world, r)[0] for r in range(world)]
    dl = d1 - d0
    if gather and world > 1 and int(pipeline) > 1 and len(set(counts)) == 1 and dl >= 2:
        nsub = min(int(pipeline), dl)
        out, pending = None, []
        for s in range(nsub):
            s0, s1 = shard_bounds(dl, nsub, s)
            loc = fn(local_vol[s0:s1], xcenter, ycenter, list_fact, row_start, nrows, coord_round_f32, blend)
            if not torch.is_tensor(loc):
                loc = torch.from_numpy(np.ascontiguousarray(loc))
            loc = loc.contiguous()
            if out is None:
                out = torch.empty((depth, nrows, loc.shape[2]), dtype=loc.dtype, device=loc.device)
            pieces = [_wire(out[r * dl + s0:r * dl + s1]) for r in range(world)]      # contiguous views: depth is the outer axis
            pending.append((dist.all_gather(pieces, _wire(loc), group=group, async_op=True), loc))
        for work, _keep in pending:
            work.wait()
        return out
    local = fn(local_vol, xcenter, ycenter, list_fact, row_start, nrows, coord_round_f32, blend)
    if not torch.is_tensor(local):
        local = torch.from_numpy(np.ascontiguousarray(local))
    if not gather or world == 1:
        return local
    width = local.shape[2]
    if len(set(counts)) == 1:
        out = torch.empty((depth, nrows, width), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(_wire(out), _wire(local.contiguous()), group=group)
        return out
    # ragged shards: pad to the largest, gather, trim
    m = max(counts)
    pad = torch.zeros((m, nrows, width), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    buf = torch.empty((world * m, nrows, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(_wire(buf), _wire(pad), group=group)
    return torch.cat([buf[r * m:r * m + counts[r]] for r in range(world)], dim=0)


def unwarp_stack_row_sharded(vol, xcenter, ycenter, list_fact, row_start, nrows, *, coord_round_f32=True, group=None,
                             blend=None, compute=None, world_size=None, rank=None):
    """
    The sharding that needs NO collective ("alternative"): every rank holds the whole stack (or
    reads it from shared storage) and owns output rows ``[r0, r1)`` of EVERY projection -- complete sinograms for its
    rows, which is what a per-sinogram reconstructor downstream consumes.  Only the source rows those output rows can
    reach are touched (the reference's band, ``postprocessing.py:289-301``; the library computes it per call).

    Returns ``(local, (r0, r1))`` with ``local`` of shape ``(depth, r1 - r0, width)``; rows are relative to ``row_start``.

    With ``coord_round_f32=True`` every rank's call is an ``unwarp_chunk_slices_backward`` over ITS rows, so the reference's
    row band (``:289-301``: spanned by the call's first and last rows) is that of the sub-chunk -- what calling the
    reference on the sub-chunk gives.  Under a model whose row coordinate increases with the row (every usable calibration)
    the band is never left and the union of the ranks' rows equals the single call; under a FOLDING model the single
    call reflects inside the whole chunk's band and the two differ, as they do in the reference.
    """
    if world_size is not None and rank is None:
        raise ValueError("rank is required when world_size is given")
    if world_size is None:
        import torch.distributed as dist
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    r0, r1 = row_shard_bounds(nrows, world_size, rank)
    fn = _hip_rows if compute is None else compute
    local = fn(vol, xcenter, ycenter, list_fact, row_start + r0, r1 - r0, coord_round_f32, blend)
    return local, (r0, r1)


def unwarp_stack_peer_gather(shards, outs, depth, height, width, xcenter, ycenter, list_fact, row_start, nrows, devices, *,
                             coord_round_f32=True, gather=True, blend=None):
    """
    Depth shards resident on the GPUs of THIS process and the exchange by direct peer copies (``hipMemcpyPeerAsync``),
    through ``dcp_unwarp_stack_rows_peer_f32`` -- no torch, no RCCL.  ``shards[g]`` / ``outs[g]``: device pointers (int) on
    ``devices[g]``; ``outs[g]`` holds ``(depth, nrows, width)`` floats when gathering.  See include/discorpy_hip.h.
    """
    import ctypes as C
    from . import _ffi as F
    from .post import postprocessing as pp
    n = len(devices)
    if len(shards) != n or len(outs) != n:
        raise ValueError("one shard and one result pointer per device slot")
    fa, nf = F.fact_array(list_fact)
    F.require_device()
    F.check(F.lib().dcp_unwarp_stack_rows_peer_f32((C.c_void_p * n)(*[int(p) if p else None for p in shards]),
                                                   (C.c_void_p * n)(*[int(p) for p in outs]), int(depth), int(height), int(width),
                                                   int(height) * int(width), int(width), float(xcenter), float(ycenter), fa, nf,
                                                   float(row_start), int(nrows), int(bool(coord_round_f32)), pp._blend_code(blend),
                                                   (C.c_int * n)(*[int(d) for d in devices]), n, int(bool(gather))))
