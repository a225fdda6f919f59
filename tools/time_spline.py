"""Kernel timings of the spline path (order 3) on cfg2-sized frames, device-resident."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
src = F.DeviceBuffer(img.nbytes).upload(img); dst = F.DeviceBuffer(img.nbytes)
fa, n = F.fact_array(c["list_fact"])
for order in (3, 2, 5):
    for mode in (0, 4):
        def run():
            F.check(L.dcp_unwarp_image_spline_f32(src.ptr, dst.ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, mode, 1, -1, None))
        run(); F.check(L.dcp_stream_synchronize(-1, None))
        e0, e1 = F.Event(), F.Event(); e0.record()
        for _ in range(5): run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        print("order %d mode %d: %.3f ms per 4096^2 frame  (%.0f Mpix/s)" % (order, mode, ms, H * W / ms / 1e3), flush=True)
