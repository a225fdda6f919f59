"""Kernel-only time of one cfg4-geometry depth shard (D projections of 2560^2, every row), float32 and uint16:
python tools/time_stack_k.py [D] [key=value options]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
D = 64
tag = ""
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); F.set_option(k, int(v)); tag += kv + " "
    else:
        D = int(kv)
c = configs.cfg4(D); _, H, W = c["shape"]
fa, n = F.fact_array(c["list_fact"])
rng = np.random.default_rng(3)
for name in ("float32", "uint16"):
    dt = np.dtype(name); es = dt.itemsize
    chunk = rng.random((4, H, W), dtype=np.float32) if name == "float32" else (rng.random((4, H, W)) * 60000).astype(dt)
    vol = F.DeviceBuffer(D * H * W * es); out = F.DeviceBuffer(D * H * W * es)
    for d in range(0, D, 4):
        F.check(L.dcp_memcpy(vol.ptr + d * H * W * es, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, -1, None))
    def run():
        if name == "float32":
            F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr, D, H, W, H * W, W, c["xcenter"], c["ycenter"], fa, n, 0.0, H, 1, F.BLEND_F64LERP, F.MEM_DEVICE, -1, None))
        else:
            F.check(L.dcp_unwarp_stack_rows_typed(vol.ptr, out.ptr, F.DTYPE_BY_NAME[name], 0, D, H, W, H * W, W, c["xcenter"], c["ycenter"], fa, n, 0.0, H, 1, F.MEM_DEVICE, -1, None))
    for _ in range(6):
        run()
    F.check(L.dcp_stream_synchronize(-1, None))
    F.debug_counters()
    e0, e1 = F.Event(), F.Event(); e0.record()
    for _ in range(10):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_ms(e1) / 10 * 1e3
    print("%-18s %-8s D=%d all rows: %8.1f us  %.3f of 8 TB/s  %s  fallbacks %s" % (tag, name, D, us, 2.0 * es * D * H * W / us / 1e6 / 8, F.last_kernel(), F.debug_counters()), flush=True)
    vol.free(); out.free()
