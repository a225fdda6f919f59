#!/usr/bin/env python
"""bench.py -- Mpixels/s of the backward unwarp on MI355X (BASELINE.json metric).

Workload at every N: BASELINE config 2 -- 4096x4096 float32 frames, 5-term backward polynomial
(coef_dot_05 rescaled), bilinear -- device-resident.  One "step" is one pass of the hot path over
a batch of `--batch` DISTINCT frames (default 24: 3.2 GB of input+output per GPU, so the 256 MiB
Infinity Cache cannot hold the working set and the kernel streams from HBM).  With N > 1 every
rank unwarps its own batch (independent frames, no data-path collective): weak scaling.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the
dominant kernel and `cpu_baseline` (oracle/unwarp_oracle.c timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402

BLEND_NAMES = {"scipy": F.BLEND_SCIPY, "f64lerp": F.BLEND_F64LERP, "f32lerp": F.BLEND_F32LERP}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pipeline", type=int, default=1,
                    help="stack workload: all-gather pipelined against the kernel in this many depth sub-blocks")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="back-to-back launches for this long before the warm-up steps: the core clock needs ~100 ms under "
                         "load to reach its sustained level (tools/time_ramp.py); 0 disables")
    ap.add_argument("--batch", type=int, default=24, help="distinct frames per step (ring size)")
    ap.add_argument("--blend", default="f64lerp", choices=sorted(BLEND_NAMES))
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the CPUs this process may use (affinity, cgroup quota)")
    ap.add_argument("--option", action="append", default=[], help="kernel option key=value (dcp_set_option)")
    ap.add_argument("--workload", default="frame", choices=["frame", "stack"],
                    help="frame: BASELINE config 2 (default, the headline metric); stack: config 4, the "
                         "(depth, 2560, 2560) stack sharded over the ranks by depth + all-gather")
    ap.add_argument("--depth", type=int, default=2048, help="stack workload: total number of projections")
    ap.add_argument("--rows", type=int, default=2560, help="stack workload: output rows per step (2560 = whole stack)")
    ap.add_argument("--no-gather", action="store_true", help="stack workload: skip the all-gather")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may really use: the smaller of the affinity mask and the cgroup quota (the GPU boxes
    show 256 logical CPUs but grant 16; running 64 threads there is slower than running 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            quota = None
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def cpu_baseline(cfg, img, blend, threads):
    """Time the oracle (a C port of the reference arithmetic) on the host: whole 4096^2 frames."""
    from oracle import oracle as orc
    ncores = orc.max_threads()
    t = threads if threads > 0 else min(ncores, usable_cpus())
    orc.set_threads(t)
    kw = dict(order=cfg["order"], poly=orc.POLY_NUMPY, blend=orc.BLEND_SCIPY)
    orc.unwarp_image_backward(img, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], **kw)  # warm
    frames, t0 = 0, time.perf_counter()
    while True:
        orc.unwarp_image_backward(img, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], **kw)
        frames += 1
        dt = time.perf_counter() - t0
        if dt * t >= 20.0 or frames >= 64:   # ~20 CPU-seconds of work
            break
    mpix = frames * img.size / dt / 1e6
    return {"value": round(mpix, 2), "unit": "Mpixels/s", "cores": t, "kind": "port",
            "sample": "%d full %dx%d frames of the bench workload, reference arithmetic order "
                      "(numpy-order polynomial, scipy blend), %d OpenMP threads (%d logical CPUs visible, %d usable "
                      "under the affinity mask / cgroup quota), %.2f s wall"
                      % (frames, img.shape[0], img.shape[1], t, ncores, usable_cpus(), dt)}


def stack_main(a, world, rank, dev, dist, backend):
    """BASELINE config 4: strong scaling -- the stack is fixed, ranks own contiguous depth shards."""
    from discorpy_amd import stack as st
    L = F.lib()
    cfg = configs.cfg4(a.depth)
    D, H, W = cfg["shape"]
    d0, d1 = st.shard_bounds(D, world, rank)
    dl = d1 - d0
    nrows = a.rows
    fa, nf = F.fact_array(cfg["list_fact"])
    blend = BLEND_NAMES[a.blend]
    gather = world > 1 and not a.no_gather
    if gather:
        import torch
        vol_t = torch.empty((dl, H, W), dtype=torch.float32, device="cuda")
        out_t = torch.empty((dl, nrows, W), dtype=torch.float32, device="cuda")
        full_t = torch.empty((D, nrows, W), dtype=torch.float32, device="cuda")
        vol_ptr, out_ptr = vol_t.data_ptr(), out_t.data_ptr()
        stream = torch.cuda.current_stream().cuda_stream
    else:
        vol_b = F.DeviceBuffer(dl * H * W * 4, dev)
        out_b = F.DeviceBuffer(dl * nrows * W * 4, dev)
        vol_ptr, out_ptr, stream = vol_b.ptr, out_b.ptr, None
    # synthetic projections: one host chunk of noise, replicated on the device
    chunk = np.random.default_rng(cfg["seed"] + rank).random((min(dl, 16), H, W), dtype=np.float32)
    done = 0
    while done < dl:
        n = min(chunk.shape[0], dl - done)
        F.check(L.dcp_memcpy(vol_ptr + done * H * W * 4, chunk.ctypes.data, n * H * W * 4, F.COPY_H2D, dev, None))
        done += n
    uneven = len({st.shard_bounds(D, world, r)[1] - st.shard_bounds(D, world, r)[0] for r in range(world)}) > 1
    if gather and uneven:
        raise SystemExit("stack workload: depth must divide evenly over the ranks for the timed all-gather")

    nsub = max(1, min(a.pipeline, dl)) if gather else 1

    def step():
        if nsub == 1:
            F.check(L.dcp_unwarp_stack_rows_f32(vol_ptr, out_ptr, dl, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                                0.0, nrows, 1, blend, F.MEM_DEVICE, dev, stream))
            if gather:
                dist.all_gather_into_tensor(full_t, out_t)
            return
        # the all-gather of sub-block s (asynchronous) runs while the kernel of sub-block s + 1 computes; every
        # rank's piece lands directly in its place of the (depth, rows, W) result
        pending = []
        for s_ in range(nsub):
            s0, s1 = st.shard_bounds(dl, nsub, s_)
            F.check(L.dcp_unwarp_stack_rows_f32(vol_ptr + s0 * H * W * 4, out_ptr + s0 * nrows * W * 4, s1 - s0, H, W, H * W, W,
                                                cfg["xcenter"], cfg["ycenter"], fa, nf, 0.0, nrows, 1, blend, F.MEM_DEVICE, dev,
                                                stream))
            pieces = [full_t[r * dl + s0:r * dl + s1] for r in range(world)]
            pending.append(dist.all_gather(pieces, out_t[s0:s1], async_op=True))
        for work in pending:
            work.wait()

    def sync():
        F.check(L.dcp_stream_synchronize(dev, None))
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt[0])
    if rank == 0:
        vox = float(D) * nrows * W * a.steps
        ms = wall * 1e3 / a.steps
        per_gpu_bytes = configs.BYTES_PER_PIXEL * dl * nrows * W
        achieved = per_gpu_bytes / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "Mpixels/s unwarp of a (depth, 2560, 2560) stack, rows of every projection (+ all-gather)",
            "value": round(vox / wall / 1e6, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (uniform [0,1) float32 projections, device-resident)",
            "config": {"workload": cfg["name"], "depth": D, "rows": nrows, "width": W, "depth_per_gpu": dl,
                       "all_gather": bool(gather), "gather_pipeline": nsub, "blend": a.blend,
                       "parallelism": "depth-sharded, %s" % ("RCCL all-gather of the (depth, rows, W) block" if gather
                                                            else "no collective")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": configs.HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / configs.HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "stack_lds_kernel / stack_rows_kernel by launch size (per GPU, step time includes the all-gather when enabled)"}}),
            flush=True)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (single-GPU boxes): DCP_BENCH_BACKEND=gloo + DCP_BENCH_DEVICE=0 let two ranks share one GPU
    backend = os.environ.get("DCP_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("DCP_BENCH_DEVICE", local_rank))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    n_gpus = world
    if a.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)

    L = F.lib()
    F.require_device()
    for kv in a.option:
        k, v = kv.split("=")
        F.set_option(k, int(v))
    dev = dev_index if world > 1 else -1
    if a.workload == "stack":
        stack_main(a, world, rank, dev, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        return
    cfg = configs.cfg2()
    cfg["order"] = a.order
    H, W = cfg["shape"]
    blend = BLEND_NAMES[a.blend]
    rng = np.random.default_rng(cfg["seed"] + 1000 * rank)
    fa, nf = F.fact_array(cfg["list_fact"])

    # device-resident batch: `batch` distinct input frames + as many output frames
    srcs, dsts, img0 = [], [], None
    for i in range(a.batch):
        img = rng.random((H, W), dtype=np.float32)
        if i == 0:
            img0 = img
        srcs.append(F.DeviceBuffer(img.nbytes, dev).upload(img))
        dsts.append(F.DeviceBuffer(img.nbytes, dev))

    def step():
        for s, d in zip(srcs, dsts):
            rc = L.dcp_unwarp_image_f32(s.ptr, d.ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                        a.order, 1, blend, F.MEM_DEVICE, dev, None)
            if rc:
                F.check(rc)

    def sync():
        F.check(L.dcp_stream_synchronize(dev, None))
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    if a.settle_ms > 0:            # let the clocks reach their sustained level before anything is counted
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < a.settle_ms:
            step()
            sync()
    for _ in range(a.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    e0, e1 = F.Event(dev), F.Event(dev)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_ms(e1)            # HIP events on the launch stream: device time of the K steps
    if dist is not None:
        import torch
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(tt[0]), float(tt[1])

    copy_gbps = None
    if rank == 0:
        # context for roofline.frac: what a plain device-to-device copy of the same frames reaches on this box
        # (hipMemcpyAsync D2D over the ring, outside the timed region)
        try:
            nbytes = H * W * 4
            for s, d in zip(srcs, dsts):
                F.check(L.dcp_memcpy(d.ptr, s.ptr, nbytes, F.COPY_D2D, dev, None))
            c0, c1 = F.Event(dev), F.Event(dev)
            c0.record()
            for _ in range(4):
                for s, d in zip(srcs, dsts):
                    F.check(L.dcp_memcpy(d.ptr, s.ptr, nbytes, F.COPY_D2D, dev, None))
            c1.record()
            c1.synchronize()
            copy_gbps = round(2.0 * nbytes * 4 * len(srcs) / (c0.elapsed_ms(c1) * 1e-3) / 1e9, 1)
        except Exception:
            copy_gbps = None

    batched = None
    if rank == 0 and n_gpus == 1 and a.order == 1:
        # context, not the metric: the frames of a step share one calibration, so the whole batch can also go
        # through ONE launch of the stack entry point (depth = batch, all rows) -- bit-identical output, the
        # coordinates evaluated once per pixel position instead of once per frame
        try:
            nbytes = H * W * 4
            vsrc, vdst = F.DeviceBuffer(nbytes * a.batch, dev), F.DeviceBuffer(nbytes * a.batch, dev)
            for i, sbuf in enumerate(srcs):
                F.check(L.dcp_memcpy(vsrc.ptr + i * nbytes, sbuf.ptr, nbytes, F.COPY_D2D, dev, None))

            def stack_step():
                F.check(L.dcp_unwarp_stack_rows_f32(vsrc.ptr, vdst.ptr, a.batch, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"],
                                                    fa, nf, 0.0, H, 1, blend, F.MEM_DEVICE, dev, None))
            stack_step()
            b0, b1 = F.Event(dev), F.Event(dev)
            b0.record()
            for _ in range(max(4, a.steps // 4)):
                stack_step()
            b1.record()
            b1.synchronize()
            per_frame_us = b0.elapsed_ms(b1) * 1e3 / (max(4, a.steps // 4) * a.batch)
            chk = np.empty((H, W), np.float32)
            F.check(L.dcp_memcpy(chk.ctypes.data, vdst.ptr + (a.batch - 1) * nbytes, nbytes, F.COPY_D2H, dev, None))
            ref = np.empty((H, W), np.float32)
            F.check(L.dcp_unwarp_image_f32(srcs[a.batch - 1].ptr, dsts[a.batch - 1].ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"],
                                           fa, nf, a.order, 1, blend, F.MEM_DEVICE, dev, None))
            F.check(L.dcp_memcpy(ref.ctypes.data, dsts[a.batch - 1].ptr, nbytes, F.COPY_D2H, dev, None))
            batched = {"what": "the %d frames of a step in one dcp_unwarp_stack_rows_f32 launch (same calibration)" % a.batch,
                       "Mpixels_per_s": round(H * W / per_frame_us, 1), "us_per_frame": round(per_frame_us, 3),
                       "frac_of_hbm_peak": round(configs.BYTES_PER_PIXEL * H * W / (per_frame_us * 1e-6) / 1e9 / configs.HBM_PEAK_GBPS, 4),
                       "identical_to_per_frame_launches": bool(np.array_equal(chk, ref))}
            vsrc.free()
            vdst.free()
        except Exception as e:      # noqa: BLE001 -- context only, never fails the bench
            batched = {"error": repr(e)}

    overlapped = None
    if rank == 0 and n_gpus == 1:
        # context, not the metric: the same frames alternated over two streams, so that the drain of one launch
        # overlaps the ramp of the next (per-launch durations are then not meaningful; only throughput is)
        try:
            s2 = [F.Stream(dev), F.Stream(dev)]

            def step2():
                for i, (s_, d_) in enumerate(zip(srcs, dsts)):
                    rc = L.dcp_unwarp_image_f32(s_.ptr, d_.ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                                a.order, 1, blend, F.MEM_DEVICE, dev, s2[i & 1].ptr)
                    if rc:
                        F.check(rc)
            for _ in range(3):
                step2()
            for st_ in s2:
                st_.synchronize()
            t2 = time.perf_counter()
            n2 = max(10, a.steps // 4)
            for _ in range(n2):
                step2()
            for st_ in s2:
                st_.synchronize()
            dt2 = time.perf_counter() - t2
            overlapped = {"what": "the same step with its frames alternated over two HIP streams",
                          "Mpixels_per_s": round(n2 * a.batch * H * W / dt2 / 1e6, 1),
                          "us_per_frame": round(dt2 * 1e6 / (n2 * a.batch), 3)}
        except Exception as e:      # noqa: BLE001 -- context only
            overlapped = {"error": repr(e)}

    if rank == 0:
        launches = a.steps * a.batch
        pix_per_launch = H * W
        total_pix = launches * pix_per_launch * n_gpus
        value = total_pix / wall / 1e6
        launch_us = dev_ms * 1e3 / launches
        achieved = configs.BYTES_PER_PIXEL * pix_per_launch / (launch_us * 1e-6) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/s backward unwarp (4096x4096, 5-term poly, bilinear)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall * 1e3 / a.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (numpy default_rng uniform [0,1) float32 frames, device-resident)",
            "config": {"workload": cfg["name"], "frames_per_step_per_gpu": a.batch, "height": H, "width": W,
                       "nfact": nf, "order": a.order, "blend": a.blend, "coord_round_f32": True, "pixel_dtype": "f32",
                       "arithmetic": "coordinates and blend in float64 (as numpy / scipy compute them), pixels float32", "clock_settle_ms": a.settle_ms,
                       "parallelism": "independent frames per GPU (no collective)" if n_gpus > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": configs.HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / configs.HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": ("remap_lds_kernel<Radial,NF=5>" if F.get_option("lds_gather") and a.order == 1
                                    else "remap_tile_kernel<Radial,NF=5>"), "launch_us": round(launch_us, 3),
                         "algorithmic_bytes_per_launch": int(configs.BYTES_PER_PIXEL * pix_per_launch),
                         "d2d_copy_same_frames_GBps": copy_gbps},
        }
        if batched is not None:
            out["batched_same_calibration"] = batched
        if overlapped is not None:
            out["two_streams"] = overlapped
        if n_gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, img0, blend, a.cpu_threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
