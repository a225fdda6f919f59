"""Per-window launch time of the bench kernel over a long back-to-back run (clock / power ramp)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
NR = 24
src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(NR)]
dst = [F.DeviceBuffer(img.nbytes) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])
t_start = time.perf_counter()
for win in range(40):
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(2400):
        k = r % NR
        L.dcp_unwarp_image_f32(src[k].ptr, dst[k].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, 1, 1, 1, 1, -1, None)
    e1.record(); e1.synchronize()
    print("t = %5.2f s: %.2f us per launch" % (time.perf_counter() - t_start, e0.elapsed_ms(e1) / 2400 * 1e3), flush=True)
