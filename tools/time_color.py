#!/usr/bin/env python
"""A/B on one box: a device-resident 4096^2 interleaved colour image (util.unwarp_color_image_backward's kernel call) through
remap_wg_color_kernel, through the one-thread-per-pixel kernel (option wg_box=0), and as NC single-plane launches of
remap_wg_kernel on planar copies.  us per image, HIP events after 300 ms of the same launches; a ring of images larger than the
256 MB Infinity Cache.

    python tools/time_color.py [--size 4096] [--ring 6] [--reps 60] [--cases f32x3,f32x4,u8x3,u16x3]
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
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--ring", type=int, default=6)
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--cases", default="f32x3,f32x4,u8x3,u16x3,f64x1,i32x1")
    ap.add_argument("--no-generic", action="store_true")
    a = ap.parse_args()
    L = F.lib()
    F.require_device()
    dev = -1
    cfg = configs.cfg2()
    s = a.size / 4096.0
    fact = [c * s ** -i for i, c in enumerate(cfg["list_fact"])]
    xc, yc = cfg["xcenter"] * s, cfg["ycenter"] * s
    fa, nf = F.fact_array(fact)
    H = W = a.size
    dt = {"f32": ("float32", 0), "u8": ("uint8", 2), "u16": ("uint16", 4), "f64": ("float64", 1), "i32": ("int32", 7)}
    rng = np.random.default_rng(3)
    for case in a.cases.split(","):
        tname, nc = case.split("x")
        nc = int(nc)
        npdt, code = dt[tname]
        es = np.dtype(npdt).itemsize
        nbytes = H * W * nc * es
        ring = max(2, min(a.ring, int(3e9 // (2 * nbytes)) or 2))
        img = (rng.random((H, W, nc), dtype=np.float32) * (255 if tname not in ("f32", "f64") else 1)).astype(npdt)
        srcs = [F.DeviceBuffer(nbytes, dev).upload(img) for _ in range(ring)]
        dsts = [F.DeviceBuffer(nbytes, dev) for _ in range(ring)]
        for name, order, blend in (("f64lerp", 1, F.BLEND_F64LERP), ("scipy", 1, F.BLEND_SCIPY), ("nearest", 0, F.BLEND_SCIPY)):
            if tname != "f32" and name == "f64lerp":
                continue

            if nc == 1:        # single planes go through the typed image entry point
                def run(i):
                    F.check(L.dcp_unwarp_image_typed(srcs[i % ring].ptr, dsts[i % ring].ptr, code, H, W, W, 1, xc, yc, fa, nf, order, 0, F.MEM_DEVICE, dev, None))
                t = bench.timed_launches(run, a.reps, dev, settle_ms=300.0)
                k = F.last_kernel()
                F.set_option("x_wg_box", 0)
                tg = bench.timed_launches(run, max(4, a.reps // 4), dev, settle_ms=100.0)
                F.set_option("x_wg_box", 1)
                print("%-6s %-8s %8.2f us  %.3f of 8 TB/s (%d B/px)  %s   | one thread per pixel: %.2f us" % (
                    case, name, t, 2 * nbytes / (t * 1e-6) / 8e12, 2 * es, k, tg), flush=True)
                continue

            def run(i):
                F.check(L.dcp_unwarp_color_image(srcs[i % ring].ptr, dsts[i % ring].ptr, code, H, W, nc, W * nc, nc, xc, yc, fa, nf, order, blend,
                                                 F.MEM_DEVICE, dev, None))
            t = bench.timed_launches(run, a.reps, dev, settle_ms=300.0)
            k = F.last_kernel()
            line = "%-6s %-8s %8.2f us  %.3f of 8 TB/s (%d B/px)  %s" % (case, name, t, 2 * nbytes / (t * 1e-6) / 8e12, 2 * nc * es, k)
            if not a.no_generic:
                F.set_option("x_wg_box", 0)
                tg = bench.timed_launches(run, max(4, a.reps // 4), dev, settle_ms=100.0)
                F.set_option("x_wg_box", 1)
                line += "   | one thread per pixel: %.2f us" % tg
            print(line, flush=True)
        if tname == "f32":
            # the same bytes as NC planar single-plane launches (what three K1 calls cost)
            planes = [F.DeviceBuffer(H * W * 4, dev).upload(np.ascontiguousarray(img[:, :, c % nc])) for c in range(nc * ring)]
            outs = [F.DeviceBuffer(H * W * 4, dev) for _ in range(nc * ring)]

            def k1(i):
                for c in range(nc):
                    j = (i % ring) * nc + c
                    F.check(L.dcp_unwarp_image_f32(planes[j].ptr, outs[j].ptr, H, W, W, 1, xc, yc, fa, nf, 1, 1, F.BLEND_F64LERP, F.MEM_DEVICE, dev, None))
            t = bench.timed_launches(k1, a.reps, dev, settle_ms=300.0)
            print("%-6s %d planar launches of %s: %.2f us" % (case, nc, F.last_kernel(), t), flush=True)
            for b in planes + outs:
                b.free()
        for b in srcs + dsts:
            b.free()


if __name__ == "__main__":
    main()
