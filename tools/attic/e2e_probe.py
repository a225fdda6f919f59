#!/usr/bin/env python
"""NumPy in -> NumPy out per 4096^2 frame (perspective and fused too), with the direct-write host path on / off and under the
runtime in use (DISCORPY_AMD_SYSTEM_HIP=1 for /opt/rocm's): min / median ms of 15 calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import configs, _ffi as F  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402

c, c3 = configs.cfg2(), configs.cfg3()
img = np.random.default_rng(1).random(c["shape"], dtype=np.float32)


def stat(fn, n=15):
    fn()
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3

F.lib()
F.require_device()
rt = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][:1]
print("runtime", rt)
ref = None
for direct in (1, 0, 2):
    F.set_option("host_direct", direct)
    r = pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"])
    if ref is None:
        ref = r.copy()
    print("host_direct %d  radial   min %.2f median %.2f ms   identical %s" % ((direct,) + stat(lambda: pp.unwarp_image_backward(
        img, c["xcenter"], c["ycenter"], c["list_fact"])) + (bool(np.array_equal(r, ref)),)), flush=True)
    print("host_direct %d  persp    min %.2f median %.2f ms" % ((direct,) + stat(lambda: pp.correct_perspective_image(img, c3["list_coef"]))), flush=True)
    print("host_direct %d  fused    min %.2f median %.2f ms" % ((direct,) + stat(lambda: pp.unwarp_perspective_fused(
        img, c3["xcenter"], c3["ycenter"], c3["list_fact"], c3["list_coef"]))), flush=True)
    print("host_direct %d  order 0  min %.2f median %.2f ms" % ((direct,) + stat(lambda: pp.unwarp_image_backward(
        img, c["xcenter"], c["ycenter"], c["list_fact"], order=0))), flush=True)
