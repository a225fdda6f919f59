"""cfg5 (8192^2, 9 terms) per-launch time, settled clocks: python tools/time_cfg5.py [key=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
tag = ""
for kv in sys.argv[1:]:
    k, v = kv.split("="); F.set_option(k, int(v)); tag += kv + " "
c = configs.cfg5(); H, W = c["shape"]
fa, n = F.fact_array(c["list_fact"])
img = np.random.default_rng(c["seed"]).random((H, W), dtype=np.float32)
src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(4)]
dst = [F.DeviceBuffer(img.nbytes) for _ in range(4)]
for order, blend in ((1, 1), (0, 0), (1, 1)):
    def run(i):
        F.check(L.dcp_unwarp_image_f32(src[i % 4].ptr, dst[i % 4].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.3:
        run(i); i += 1
        if i % 16 == 0:
            F.check(L.dcp_stream_synchronize(-1, None))
    F.check(L.dcp_stream_synchronize(-1, None))
    F.debug_counters()
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(200):
        run(r)
    e1.record(); e1.synchronize()
    us = e0.elapsed_ms(e1) / 200 * 1e3
    print("%-22s order %d: %7.2f us %.3f of 8 TB/s  %s  fallbacks/launch %s" % (tag, order, us, 8.0 * H * W / us / 1e6 / 8, F.last_kernel(), [v / 200 for v in F.debug_counters()]), flush=True)
