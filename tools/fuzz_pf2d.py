#!/usr/bin/env python
"""Randomised cases aimed at spline_prefilter2d_kernel: float32 / uint8 / uint16 / int16 frames with lines long enough for the one-pass kernels, orders 2 .. 5
(the two-pole orders 4 / 5 as two passes of the kernel, one pole each; FUZZ_ORDERS=2,3 restricts them),
every boundary mode, row-padded and channel-strided views, radial and perspective maps, random rows per chunk -- against the oracle
(<= 1 float32 ulp on <= 8 pixels, the criterion of tests/test_spline_prefilter2d.py; results that cancel to nearly zero against
the scale of the data).

    python tools/fuzz_pf2d.py [cases] [seed]
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DTYPES = ("float32", "float32", "uint8", "uint16", "int16")
MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "wrap", "grid-wrap")


def ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


ORDERS = [int(v) for v in os.environ.get("FUZZ_ORDERS", "2,3,4,5").split(",")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    orc.build()
    orc.set_threads(min(32, orc.max_threads()))
    F.lib()
    F.require_device()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    fused = 0
    by_type = {}
    for k in range(n):
        h, w = int(rng.integers(566, 2400)), int(rng.integers(566, 2400))
        if rng.integers(0, 4) == 0:
            h, w = (int(rng.integers(566, 700)), int(rng.integers(3000, 7000))) if rng.integers(0, 2) else (int(rng.integers(3000, 7000)), int(rng.integers(566, 700)))
        order = int(rng.choice(ORDERS))
        if order >= 4 and rng.integers(0, 5):          # z1^n underflows from 731 (order 4) / 884 (order 5) samples on: mostly lines that long
            h, w = max(h, int(rng.integers(890, 1500))), max(w, int(rng.integers(890, 1500)))
        mode = MODES[int(rng.integers(0, len(MODES)))]
        layout = int(rng.integers(0, 4))
        dt = np.dtype(DTYPES[int(rng.integers(0, len(DTYPES)))])
        if dt.kind == "f":
            dense = rng.random((h, w), dtype=np.float32) * np.float32(rng.choice([1.0, 255.0, 65535.0])) - np.float32(rng.choice([0.0, 100.0]))
        else:
            info = np.iinfo(dt)
            dense = rng.integers(info.min, info.max, size=(h, w), endpoint=True, dtype=np.int64).astype(dt)
        if layout == 1:                       # row-padded view
            img = np.zeros((h, w + int(rng.integers(1, 70))), dt)[:, :w]
        elif layout == 2:                     # one channel of an interleaved image
            c = int(rng.integers(2, 5))
            img = np.zeros((h, w, c), dt)[:, :, int(rng.integers(0, c))]
        elif layout == 3:                     # a band of rows and columns of a larger frame
            img = np.zeros((h + 40, w + 50), dt)[17:17 + h, 23:23 + w]
        else:
            img = np.zeros((h, w), dt)
        img[...] = dense
        F.set_option("x_pf2d_chunk", int(rng.choice([0, 0, 64, 96, 160, 320])))
        F.set_option("x_pf2d_xcd", int(rng.integers(0, 2)))
        tag = "case %d %dx%d %s order %d mode %s layout %d strides %r" % (k, h, w, dt.name, order, mode, layout, img.strides)
        if rng.integers(0, 3):
            xc, yc = float(rng.uniform(0.2, 0.8) * w), float(rng.uniform(0.2, 0.8) * h)
            fact = [1.0] + [float(rng.uniform(-0.04, 0.04)) / float(np.hypot(h, w)) ** i for i in range(1, int(rng.integers(2, 5)))]
            got = pp.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode)
            name = F.last_kernel()
            want = orc.unwarp_image_backward(dense, xc, yc, fact, order=order, mode=mode, poly=orc.POLY_KERNEL)
        else:
            coef = [1.0 + rng.uniform(-0.03, 0.03), rng.uniform(-0.02, 0.02), rng.uniform(-9, 9), rng.uniform(-0.02, 0.02), 1.0 + rng.uniform(-0.03, 0.03),
                    rng.uniform(-9, 9), rng.uniform(-3e-6, 3e-6), rng.uniform(-3e-6, 3e-6)]
            got = pp.correct_perspective_image(img, coef, order=order, mode=mode)
            name = F.last_kernel()
            want = orc.correct_perspective_image(dense, coef, order=order, mode=mode)
        fused += "prefilter2d" in name
        by_type[dt.name] = by_type.get(dt.name, 0) + ("prefilter2d" in name)
        assert got.dtype == dt and want.dtype == dt, (tag, got.dtype, want.dtype)
        if dt.kind != "f":
            # (integer results: scipy's rounding of a float64 sum that may sit on a half -- tools/fuzz_parity.py's criterion)
            di = np.abs(got.astype(np.int64) - want.astype(np.int64))
            assert di.max() <= 1 and np.count_nonzero(di) <= max(3, got.size // 2000), (tag, name, int(di.max()), int(np.count_nonzero(di)))
            continue
        d = ulps(got, want)
        # (a result that is nearly zero by cancellation -- the data cross zero in a third of the cases -- carries the float64 noise of
        # the factorised tap sum at many of ITS ulps: judged against the scale of the data there, 2^-40 of it)
        far = (d > 1) & (np.abs(got.astype(np.float64) - want) > 2.0 ** -40 * float(np.max(np.abs(want))))
        assert not far.any() and np.count_nonzero(d) <= 8, (tag, name, int(d.max()), int(np.count_nonzero(d)))
    F.set_option("x_pf2d_chunk", 0)
    F.set_option("x_pf2d_xcd", 1)
    print("fuzz_pf2d: %d cases (seed %d) within one float32 ulp on <= 8 pixels of the oracle in %.1f s; spline_prefilter2d_kernel ran in %d %r" % (
        n, seed, time.time() - t0, fused, by_type))


if __name__ == "__main__":
    main()
