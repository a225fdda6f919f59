#!/usr/bin/env python
"""Same-box A/B of the integer blend: option int_exact = 1 (the factorised blend where it is provably exact) against 0 (scipy's
operation order everywhere), uint16 / uint8 4096^2 frames and a uint16 64-projection shard of config 4; the two modes must give
identical outputs.  us per launch (HIP events after 200 ms of the same launches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402

L = F.lib()
F.require_device()
dev = -1
c2 = configs.cfg2()
H, W = c2["shape"]
fa, nf = F.fact_array(c2["list_fact"])
rng = np.random.default_rng(3)
for name, dt, scale in (("uint16", np.uint16, 60000.0), ("uint8", np.uint8, 255.0), ("int16", np.int16, 30000.0)):
    img = (rng.random((H, W)) * scale).astype(dt)
    code = F.DTYPE_BY_NAME[name]
    src = [F.DeviceBuffer(img.nbytes, dev).upload(img) for _ in range(8)]
    dst = [F.DeviceBuffer(img.nbytes, dev) for _ in range(8)]

    def run(i):
        F.check(L.dcp_unwarp_image_typed(src[i % 8].ptr, dst[i % 8].ptr, code, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf, 1, 0, F.MEM_DEVICE, dev, None))
    outs, ts = {}, {0: [], 1: []}
    for rep in range(3):
        for mode in (1, 0):
            F.set_option("x_int_exact", mode)
            ts[mode].append(bench.timed_launches(run, 400, dev, settle_ms=200.0))
            run(0)
            outs[mode] = dst[0].download((H, W), dt)
    F.set_option("x_int_exact", 1)
    print("%-7s frame 4096^2: exact %s us   scipy-order %s us   identical %s   (%s)" % (
        name, ["%.2f" % t for t in ts[1]], ["%.2f" % t for t in ts[0]], bool(np.array_equal(outs[0], outs[1])), F.last_kernel()), flush=True)
    for b in src + dst:
        b.free()

c4 = configs.cfg4(64)
D, Hs, Ws = c4["shape"]
f4, n4 = F.fact_array(c4["list_fact"])
chunk = (rng.random((4, Hs, Ws)) * 60000.0).astype(np.uint16)
vol = F.DeviceBuffer(D * Hs * Ws * 2, dev)
out = F.DeviceBuffer(D * Hs * Ws * 2, dev)
for d in range(0, D, 4):
    F.check(L.dcp_memcpy(vol.ptr + d * Hs * Ws * 2, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, dev, None))


def shard(_i):
    F.check(L.dcp_unwarp_stack_rows_typed(vol.ptr, out.ptr, F.DTYPE_BY_NAME["uint16"], 0, D, Hs, Ws, Hs * Ws, Ws, c4["xcenter"], c4["ycenter"], f4, n4, 0.0, Hs, 1,
                                          F.MEM_DEVICE, dev, None))
outs, ts = {}, {0: [], 1: []}
for rep in range(3):
    for mode in (1, 0):
        F.set_option("x_int_exact", mode)
        ts[mode].append(bench.timed_launches(shard, 12, dev, settle_ms=100.0))
        g = np.empty((2, Hs, Ws), np.uint16)
        F.check(L.dcp_memcpy(g.ctypes.data, out.ptr + 5 * Hs * Ws * 2, g.nbytes, F.COPY_D2H, dev, None))
        outs[mode] = g
F.set_option("x_int_exact", 1)
print("uint16 shard (64, 2560, 2560): exact %s us   scipy-order %s us   identical %s   (%s)" % (
    ["%.1f" % t for t in ts[1]], ["%.1f" % t for t in ts[0]], bool(np.array_equal(outs[0], outs[1])), F.last_kernel()), flush=True)
