"""Run the cfg2 frame kernel back to back for argv[1] seconds (blend argv[2], order argv[3]) and print the sustained per-launch time --
the load under which tools/clock_probe.sh samples clocks and power.  DCP_LIB_PATH selects the build."""
import sys
import time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
secs, blend, order = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = configs.cfg2(); H, W = c["shape"]
rng = np.random.default_rng(c["seed"])
NR = 24
NLAUNCH = 4800 if secs > 0 else 96
src = [F.DeviceBuffer(H * W * 4).upload(rng.random((H, W), dtype=np.float32)) for _ in range(NR)]
dst = [F.DeviceBuffer(H * W * 4) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])
t0 = time.perf_counter(); launches = 0
while True:
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(NLAUNCH):
        F.check(L.dcp_unwarp_image_f32(src[r % NR].ptr, dst[r % NR].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    e1.record(); e1.synchronize()
    us = e0.elapsed_ms(e1) / NLAUNCH * 1e3
    if time.perf_counter() - t0 >= secs:
        break
print("%.2f us per launch (last 4800)" % us, flush=True)
