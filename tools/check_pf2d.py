#!/usr/bin/env python
"""The one-launch 2-D prefilter (spline_prefilter2d_kernel, option x_spline_tiled = 1, the default) against the two-launch prefilter (= 6)
and the oracle on frames with partial stripes / chunks, both one-pole orders and both boundary kinds; then us per 4096^2 frame of
each, and of the fused kernel by rows per chunk.

    python tools/check_pf2d.py [--chunks 0,64,128,192,256,384,512]
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
from oracle import oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", default="0,64,128,192,256,384,512")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--fast", type=int, default=1)
    ap.add_argument("--slow", type=int, default=6)
    ap.add_argument("--xcd", default="1")                      # x_pf2d_xcd values to time (0: plain launch order)
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-timing", action="store_true")
    a = ap.parse_args()
    orc.build()
    orc.set_threads(min(32, orc.max_threads()))
    L = F.lib()
    F.require_device()
    c = configs.cfg2()
    bad = 0
    for shape in () if a.skip_parity else ((1100, 1347), (600, 2100), (2100, 700), (1024, 4096), (2000, 1500), (569, 571), (566, 6000), (7000, 566)):
        img = np.random.default_rng(5).random(shape, dtype=np.float32)
        for order, mode in [(3, "reflect"), (3, "mirror"), (2, "reflect"), (2, "mirror"), (3, "grid-mirror"), (3, "nearest"), (2, "grid-constant"), (3, "grid-constant")]:
            args = (img, c["xcenter"] * shape[1] / 4096, 0.45 * shape[0], c["list_fact"])
            want = orc.unwarp_image_backward(*args, order=order, mode=mode, poly=orc.POLY_KERNEL)
            res = {}
            names = {}
            for t in (a.slow, a.fast):
                F.set_option("x_spline_tiled", t)
                for ch in ((0,) if t == a.slow else (0, 64, 96)):
                    F.set_option("x_pf2d_chunk", ch)
                    res[(t, ch)] = pp.unwarp_image_backward(*args, order=order, mode=mode)
                    names[(t, ch)] = F.last_kernel()
            F.set_option("x_pf2d_chunk", 0)
            for key in sorted(res):
                if key[0] == a.slow:
                    continue
                d_or = int(np.count_nonzero(res[key] != want))
                d_ab = int(np.count_nonzero(res[key] != res[(a.slow, 0)]))
                mx = float(np.max(np.abs(res[key].astype(np.float64) - want)))
                ok = d_or <= 8 and d_ab <= 4 and mx < 1e-5 and "prefilter2d" in names[key]
                bad += not ok
                print(shape, order, mode, "chunk", key[1], "!= oracle: %d px, != two-launch: %d px, max |d| %.3g" % (d_or, d_ab, mx), names[key],
                      "OK" if ok else "BAD", flush=True)
    print("bad", bad, flush=True)
    F.set_option("x_spline_tiled", 1)
    if a.skip_timing:
        sys.exit(1 if bad else 0)
    # ---- timing
    dev = -1
    H, W = c["shape"]
    fa, nf = F.fact_array(c["list_fact"])
    rng = np.random.default_rng(2)
    ring = 8
    srcs = [F.DeviceBuffer(H * W * 4, dev).upload(rng.random((H, W), dtype=np.float32)) for _ in range(ring)]
    dsts = [F.DeviceBuffer(H * W * 4, dev) for _ in range(ring)]
    for order in (3, 2):
        def run(i):
            F.check(L.dcp_unwarp_image_spline_f32(srcs[i % ring].ptr, dsts[i % ring].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, order, 0,
                                                  F.MEM_DEVICE, dev, None))
        for rep in range(2):
            F.set_option("x_spline_tiled", a.slow)
            t = bench.timed_launches(run, a.reps, dev, settle_ms=300.0)
            print("order %d two-launch prefilter: %8.2f us  %s" % (order, t, F.last_kernel()), flush=True)
            F.set_option("x_spline_tiled", a.fast)
            for xcd in [int(v) for v in a.xcd.split(",")]:
                F.set_option("x_pf2d_xcd", xcd)
                for ch in [int(v) for v in a.chunks.split(",")]:
                    F.set_option("x_pf2d_chunk", ch)
                    t = bench.timed_launches(run, a.reps, dev, settle_ms=300.0)
                    print("order %d fused prefilter, xcd order %d, chunk %4d: %8.2f us  %s" % (order, xcd, ch, t, F.last_kernel()), flush=True)
            F.set_option("x_pf2d_chunk", 0)
            F.set_option("x_pf2d_xcd", 1)
    F.set_option("x_spline_tiled", 1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
