"""Sustained per-launch time of the cfg2 frame kernel (HIP events, ring of 24 frame pairs, 300 ms of launches
first so the clock has settled).  DCP_LIB_PATH selects the build; argv: tag [key=value options].
    DCP_LIB_PATH=discorpy_amd/lib/libdcp_r01.so python tools/time_k1.py r01"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for kv in sys.argv[2:]:
    k, v = kv.split("="); F.set_option(k, int(v)); tag += " " + kv
c = configs.cfg2(); H, W = c["shape"]
rng = np.random.default_rng(c["seed"])
NR = 24
src = [F.DeviceBuffer(H * W * 4).upload(rng.random((H, W), dtype=np.float32)) for _ in range(NR)]
dst = [F.DeviceBuffer(H * W * 4) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])


def launch(k, order, blend):
    F.check(L.dcp_unwarp_image_f32(src[k].ptr, dst[k].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))


def run(order, blend, reps=1920):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for k in range(NR):
            launch(k, order, blend)
        F.check(L.dcp_stream_synchronize(-1, None))
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(reps):
        launch(r % NR, order, blend)
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


for name, order, blend in [("f64lerp", 1, 1), ("scipy", 1, 0), ("f32lerp", 1, 2), ("nearest", 0, 0), ("f64lerp", 1, 1)]:
    us = run(order, blend)
    print("%-24s %-8s %7.2f us  %5.3f of 8 TB/s  %s" % (tag, name, us, 8.0 * H * W / us / 1e6 / 8.0, F.last_kernel()), flush=True)
