"""Kernel-only (HIP events, device-resident ring) per-launch times of the element-type entry point on 4096^2 frames:
python tools/time_typed_k.py [key=value options]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
for kv in sys.argv[1:]:
    k, v = kv.split("="); F.set_option(k, int(v))
c = configs.cfg2(); H, W = c["shape"]
fa, n = F.fact_array(c["list_fact"])
rng = np.random.default_rng(5)
for name in ("uint16", "uint8", "int16", "float32"):
    dt = np.dtype(name); code = F.DTYPE_BY_NAME[name]
    NR = 16
    img = (rng.random((H, W)) * (200 if dt.itemsize == 1 else 60000)).astype(dt) if dt.kind != "f" else rng.random((H, W), dtype=np.float32)
    src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(NR)]
    dst = [F.DeviceBuffer(img.nbytes) for _ in range(NR)]
    for order in (1, 0):
        def run(i):
            F.check(L.dcp_unwarp_image_typed(src[i % NR].ptr, dst[i % NR].ptr, code, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 0,
                                             F.MEM_DEVICE, -1, None))
        t0 = time.perf_counter(); i = 0
        while time.perf_counter() - t0 < 0.25:
            run(i); i += 1
            if i % 64 == 0:
                F.check(L.dcp_stream_synchronize(-1, None))
        F.check(L.dcp_stream_synchronize(-1, None))
        e0, e1 = F.Event(), F.Event(); e0.record()
        for r in range(480):
            run(r)
        e1.record(); e1.synchronize()
        us = e0.elapsed_ms(e1) / 480 * 1e3
        bpp = 2 * dt.itemsize
        print("%-8s order %d: %7.2f us  %5.2f TB/s algorithmic (%.3f of 8 TB/s)  %s" % (name, order, us, bpp * H * W / us / 1e6, bpp * H * W / us / 1e6 / 8, F.last_kernel()), flush=True)
    for b in src + dst:
        b.free()
