#!/usr/bin/env python
"""A/B on one box: a device-resident 4096^2 float32 frame through dcp_unwarp_image_spline_f32 (orders 2..5, mode reflect) with the
register-streaming column pass (option spline_tiled = 1) and with the LDS tile kernel on both axes (= 2).  us per frame, HIP
events after 300 ms of the same launches; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.

    python tools/time_spline.py [--orders 2,3,5] [--reps 30]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", default="2,3,5")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--variants", default="1,2,3,4")      # 1 default, 2 tile kernel both axes, 3 unstaged column stream, 4 row scan
    a = ap.parse_args()
    L = F.lib()
    F.require_device()
    dev = -1
    cfg = configs.cfg2()
    H, W = cfg["shape"]
    fa, nf = F.fact_array(cfg["list_fact"])
    rng = np.random.default_rng(2)
    ring = 8
    srcs = [F.DeviceBuffer(H * W * 4, dev).upload(rng.random((H, W), dtype=np.float32)) for _ in range(ring)]
    dsts = [F.DeviceBuffer(H * W * 4, dev) for _ in range(ring)]
    for order in [int(v) for v in a.orders.split(",")]:
        def run(i):
            F.check(L.dcp_unwarp_image_spline_f32(srcs[i % ring].ptr, dsts[i % ring].ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf, order, 0,
                                                  F.MEM_DEVICE, dev, None))
        outs = {}
        for v in [int(x) for x in a.variants.split(",")]:
            F.set_option("x_spline_tiled", v)
            t = bench.timed_launches(run, a.reps, dev, settle_ms=300.0)
            run(0)
            outs[v] = bench.download(dsts[0].ptr, (H, W), dev)
            import hashlib
            print("order %d spline_tiled=%d: %8.2f us  %s  sha %s" % (order, v, t, F.last_kernel(), hashlib.sha256(outs[v].tobytes()).hexdigest()[:12]), flush=True)
        F.set_option("x_spline_tiled", 1)
        ks = sorted(outs)
        for v in ks[1:]:
            d = outs[ks[0]] != outs[v]
            print("   pixels differing between spline_tiled=%d and %d: %d of %d (max |diff| %.3g)" % (
                ks[0], v, int(d.sum()), d.size, float(np.max(np.abs(outs[ks[0]].astype(np.float64) - outs[v]))) if d.any() else 0.0), flush=True)


if __name__ == "__main__":
    main()
