#!/usr/bin/env python
"""NumPy in -> NumPy out (PCIe included) on a 4096^2 float32 frame through post.unwarp_image_backward, under option sets of the host
paths, alternated in one process.  The front end binds to the HIP runtime bundled with PyTorch whenever torch is installed
(discorpy_amd/_ffi.py); --system sets DISCORPY_AMD_SYSTEM_HIP=1 first, which leaves the library on /opt/rocm's.

    python tools/time_host_bands.py [--system] [--reps 21] [--sets "host_bands=6;host_bands=12;host_direct=2"] [--fresh]
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
ap.add_argument("--system", action="store_true")
ap.add_argument("--reps", type=int, default=21)
ap.add_argument("--sets", default="host_bands=6;host_bands=12")
ap.add_argument("--fresh", action="store_true", help="a new source array for every call (the runtime's pin cache has not seen it)")
a = ap.parse_args()
if a.system:
    os.environ["DISCORPY_AMD_SYSTEM_HIP"] = "1"
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402

c = configs.cfg2()
rng = np.random.default_rng(1)
img = rng.random(c["shape"], dtype=np.float32)
pool = [img.copy() for _ in range(a.reps + 2)] if a.fresh else None
ref = None
sets = [s for s in a.sets.split(";") if s]
for rnd in range(2):
    for s in sets:
        kv = [x.split("=") for x in s.split(",")]
        old = {k: F.get_option(k) for k, _ in kv}
        for k, v in kv:
            F.set_option(k, int(v))
        out = pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"])
        if ref is None:
            ref = out.copy()
        same = bool(np.array_equal(out, ref))
        ts = []
        for i in range(a.reps):
            src = pool[i] if a.fresh else img
            t0 = time.perf_counter()
            out = pp.unwarp_image_backward(src, c["xcenter"], c["ycenter"], c["list_fact"])
            ts.append(time.perf_counter() - t0)
            del out
        ts.sort()
        print("%s runtime  %-28s %s  median %.3f ms  min %.3f  max %.3f  kernel %s  equal %s" % (
            "system" if a.system else "torch-bundled", s, "fresh arrays" if a.fresh else "same array  ", ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3,
            F.last_kernel(), same), flush=True)
        for k, v in old.items():
            F.set_option(k, v)
