#!/usr/bin/env python
"""A/B on one box: BASELINE config 2 frames through (a) one dcp_unwarp_image_f32 launch per frame, (b) ONE dcp_unwarp_images_f32
call for the ring (remap_wg_batch_kernel, every frame its own calibration), (c) the stack entry point (same calibration), for
every sampler.  us per 4096^2 frame, HIP events on the launch stream after 300 ms of the same launches.

    python tools/time_batch.py [--batch 24] [--reps 40] [--option key=value ...]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--samplers", default="f64lerp,scipy,f32lerp,nearest")
    a = ap.parse_args()
    L = F.lib()
    F.require_device()
    for kv in a.option:
        k, v = kv.split("=")
        F.set_option(k, int(v))
    dev = -1
    cfg = configs.cfg2()
    H, W = cfg["shape"]
    n = a.batch
    rng = np.random.default_rng(1)
    srcs = [F.DeviceBuffer(H * W * 4, dev).upload(rng.random((H, W), dtype=np.float32)) for _ in range(n)]
    dsts = [F.DeviceBuffer(H * W * 4, dev) for _ in range(n)]
    cals = bench.distinct_calibrations(cfg, n)
    nf = len(cfg["list_fact"])
    table = np.ascontiguousarray([c[2] for c in cals], dtype=np.float64)
    sp = (C.c_void_p * n)(*[b.ptr for b in srcs])
    dp = (C.c_void_p * n)(*[b.ptr for b in dsts])
    xa, ya = (C.c_double * n)(*[c[0] for c in cals]), (C.c_double * n)(*[c[1] for c in cals])
    tp = table.ctypes.data_as(C.POINTER(C.c_double))
    fa, _ = F.fact_array(cfg["list_fact"])
    for name in a.samplers.split(","):
        order = 0 if name == "nearest" else 1
        blend = F.BLEND_SCIPY if name == "nearest" else bench.BLEND_NAMES[name]

        def per_frame(i):
            F.check(L.dcp_unwarp_image_f32(srcs[i % n].ptr, dsts[i % n].ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf, order, 1, blend,
                                           F.MEM_DEVICE, dev, None))

        def batch(_i):
            F.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xa, ya, tp, nf, order, 1, blend, F.MEM_DEVICE, dev, None))
        t1 = bench.timed_launches(per_frame, a.reps * n, dev, settle_ms=300.0)
        k1 = F.last_kernel()
        t2 = bench.timed_launches(batch, a.reps, dev, settle_ms=300.0) / n
        k2 = F.last_kernel()
        t1b = bench.timed_launches(per_frame, a.reps * n, dev, settle_ms=300.0)
        print("%-8s per-launch %.2f / %.2f us (%s)   batched %.2f us per frame (%s)" % (name, t1, t1b, k1, t2, k2), flush=True)


if __name__ == "__main__":
    main()
