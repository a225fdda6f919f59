"""How often does the kernels' polynomial order land on a different float32 coordinate than
numpy's?  ("parity".)  Restates the reference expression of
discorpy/post/postprocessing.py:138-145 in numpy and compares it with the oracle's
ORC_POLY_KERNEL (fused final step, what the HIP kernels compute) and ORC_POLY_KERNEL_MULADD
(separate multiply and add) orders on 4096^2 frames with perturbed models.  CPU only.

    python tools/flip_rate.py [trials]
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import configs  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def numpy_coords(h, w, xc, yc, fact):
    xu = np.arange(w) - xc
    yu = np.arange(h) - yc
    xm, ym = np.meshgrid(xu, yu)
    ru = np.sqrt(xm ** 2 + ym ** 2)
    fm = np.sum(np.asarray([f * ru ** i for i, f in enumerate(fact)]), axis=0)
    return np.float32(np.clip(yc + fm * ym, 0, h - 1)), np.float32(np.clip(xc + fm * xm, 0, w - 1))


def main(trials):
    orc.set_threads(min(16, orc.max_threads()))
    tot = {"kernel(fma)": 0, "kernel(mul+add)": 0}
    n = 0
    for t in range(trials):
        rng = np.random.default_rng(t)
        h = w = 4096
        xc, yc, fact = configs.rescale_model(w)
        xc += rng.uniform(-300, 300)
        yc += rng.uniform(-300, 300)
        fact = [f * (1 + 0.2 * rng.uniform(-1, 1)) if i else f for i, f in enumerate(fact)]
        yr, xr = numpy_coords(h, w, xc, yc, fact)
        n += 2 * h * w
        for name, mode in (("kernel(fma)", orc.POLY_KERNEL), ("kernel(mul+add)", orc.POLY_KERNEL_MULADD)):
            yo, xo = orc.radial_coords(h, w, xc, yc, fact, poly=mode)
            tot[name] += int((xo.astype(np.float32) != xr).sum()) + int((yo.astype(np.float32) != yr).sum())
        print("trial %d: coordinates %d, differing float32 coordinates so far %s" % (t, n, tot), flush=True)
    for k, v in tot.items():
        print("%-16s %d of %d float32 coordinates differ from numpy's (%.2e)" % (k, v, n, v / n))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
