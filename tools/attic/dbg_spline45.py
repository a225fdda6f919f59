import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp
from oracle import oracle as orc
orc.build(); orc.set_threads(16)
rng = np.random.default_rng(3)
h, w, dt, order, mode, xc, yc, fact = (1571, 1532, np.float32, 4, "reflect", 499.99635858988756, 218.84785084205987, [1.0, 1.9458361635865997e-05, 6.617751424219899e-09, 8.312873297400471e-13])
for _ in range(2):
    rng.random((10, 10))
img = (np.random.default_rng(3).random((1524, 1194)) * 400.0 - 100.0)  # consume as the earlier script did
img = (np.random.default_rng(5).random((h, w)) * 400.0 - 100.0).astype(dt)
want = orc.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode, poly=orc.POLY_KERNEL)
print("bounds build:", F.debug_bounds())
for rep in range(4):
    for blend in (None, "scipy"):
        got = pp.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode, blend=blend)
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = np.argwhere(d > 1e-4)
        print(rep, "blend", blend, "max|d| %.3g" % d.max(), "n bad", len(bad), ("rows %d-%d cols %d-%d" % (bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max())) if len(bad) else "", "nonzero bad values", int((got[tuple(bad.T)] != 0).sum()) if len(bad) else 0)
print("bounds:", F.debug_bounds())
for o2 in (2, 3, 5):
    want2 = orc.unwarp_image_backward(img, xc, yc, fact, order=o2, mode=mode, poly=orc.POLY_KERNEL)
    got = pp.unwarp_image_backward(img, xc, yc, fact, order=o2, mode=mode)
    d = np.abs(got.astype(np.float64) - want2.astype(np.float64)); print("order", o2, "max|d| %.3g" % d.max(), "n bad", int((d > 1e-4).sum()))
