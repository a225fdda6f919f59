#!/usr/bin/env python
"""us per device-resident 4096^2 float32 frame through dcp_unwarp_image_spline_f32 for every spline order x boundary mode, with the
kernels each call launched (run under `rocprofv3 --kernel-trace --stats` for the split).

    python tools/time_spline_modes.py [--orders 2,3,4,5] [--modes reflect,mirror,nearest,grid-constant,constant,wrap]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", default="2,3,4,5")
    ap.add_argument("--modes", default="reflect,mirror,nearest,grid-constant,constant,wrap")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--dtype", default="float32")            # float32: dcp_unwarp_image_spline_f32; others: dcp_unwarp_image_typed
    ap.add_argument("--tiled", default="1")                  # x_spline_tiled values (1 the default, 6 the launches of rounds 3-5)
    a = ap.parse_args()
    L = F.lib()
    F.require_device()
    c = configs.cfg2()
    H, W = c["shape"]
    fa, nf = F.fact_array(c["list_fact"])
    rng = np.random.default_rng(2)
    dt = np.dtype(a.dtype)
    frames = [(rng.random((H, W), dtype=np.float32) * (1.0 if dt.kind == "f" else 60000.0 if dt.itemsize > 1 else 250.0)).astype(dt) for _ in range(4)]
    srcs = [F.DeviceBuffer(H * W * dt.itemsize, -1).upload(fr) for fr in frames]
    dsts = [F.DeviceBuffer(H * W * dt.itemsize, -1) for _ in range(4)]
    for order in [int(v) for v in a.orders.split(",")]:
        for mode in a.modes.split(","):
            m = pp._spline_mode(mode, None)

            def run(i):
                if dt == np.float32:
                    F.check(L.dcp_unwarp_image_spline_f32(srcs[i % 4].ptr, dsts[i % 4].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, order, m,
                                                          F.MEM_DEVICE, -1, None))
                else:
                    F.check(L.dcp_unwarp_image_typed(srcs[i % 4].ptr, dsts[i % 4].ptr, F.DTYPE_BY_NAME[dt.name], H, W, W, 1, c["xcenter"], c["ycenter"],
                                                     fa, nf, order, m, F.MEM_DEVICE, -1, None))
            for tiled in [int(v) for v in a.tiled.split(",")]:
                F.set_option("x_spline_tiled", tiled)
                t = bench.timed_launches(run, a.reps, -1, settle_ms=150.0)
                print("%s order %d %-14s x_spline_tiled %d %8.1f us  %s" % (dt.name, order, mode, tiled, t, F.last_kernel()), flush=True)
            F.set_option("x_spline_tiled", 1)


if __name__ == "__main__":
    main()
