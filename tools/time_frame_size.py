"""Per-launch time of the radial frame kernel at a given square size (cfg2 model rescaled): python tools/time_frame_size.py SIZE [key=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
N = int(sys.argv[1]); tag = ""
for kv in sys.argv[2:]:
    k, v = kv.split("="); F.set_option(k, int(v)); tag += kv + " "
xc, yc, fact = configs.rescale_model(N)
fa, n = F.fact_array(fact)
NR = max(4, min(24, int(3.2e9 / (8 * N * N))))
img = np.random.default_rng(1).random((N, N), dtype=np.float32)
src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(NR)]
dst = [F.DeviceBuffer(img.nbytes) for _ in range(NR)]
def run(i):
    F.check(L.dcp_unwarp_image_f32(src[i % NR].ptr, dst[i % NR].ptr, N, N, N, 1, xc, yc, fa, n, 1, 1, F.BLEND_F64LERP, 1, -1, None))
t0 = time.perf_counter(); i = 0
while time.perf_counter() - t0 < 0.3:
    run(i); i += 1
    if i % 64 == 0:
        F.check(L.dcp_stream_synchronize(-1, None))
F.check(L.dcp_stream_synchronize(-1, None))
reps = max(200, int(2e10 / (N * N)))
e0, e1 = F.Event(), F.Event(); e0.record()
for r in range(reps):
    run(r)
e1.record(); e1.synchronize()
us = e0.elapsed_ms(e1) / reps * 1e3
print("%-14s %5d^2: %8.2f us  %.3f of 8 TB/s  %s" % (tag, N, us, 8.0 * N * N / us / 1e6 / 8, F.last_kernel()), flush=True)
