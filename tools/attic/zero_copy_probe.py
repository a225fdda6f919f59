#!/usr/bin/env python
"""Can the frame kernel write its result straight into REGISTERED host memory at PCIe rate (no D2H copy), and read its source from
it?  dcp_unwarp_image_f32 with mem_kind = DEVICE and a hipHostRegister'ed NumPy array as dst (and as src)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F, configs  # noqa: E402

L = F.lib()
F.require_device()
path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
hip = C.CDLL(path)
print("runtime:", path)
c = configs.cfg2()
H, W = c["shape"]
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
out = np.zeros((H, W), np.float32)
fa, nf = F.fact_array(c["list_fact"])
dsrc = F.DeviceBuffer(img.nbytes).upload(img)
ddst = F.DeviceBuffer(img.nbytes)
assert hip.hipHostRegister(C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes), 0) == 0
assert hip.hipHostRegister(C.c_void_p(img.ctypes.data), C.c_size_t(img.nbytes), 0) == 0


def run(src, dst):
    F.check(L.dcp_unwarp_image_f32(src, dst, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, 1, 1, F.BLEND_F64LERP, F.MEM_DEVICE, -1, None))
    F.check(L.dcp_stream_synchronize(-1, None))


def best(fn, reps=7):
    fn()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3

print("device -> device                  %.3f ms" % best(lambda: run(dsrc.ptr, ddst.ptr)))
print("device -> registered host (write) %.3f ms" % best(lambda: run(dsrc.ptr, out.ctypes.data)))
ref = ddst.download((H, W), np.float32)
print("   identical to the device result:", bool(np.array_equal(ref, out)))
print("registered host (read) -> device  %.3f ms" % best(lambda: run(img.ctypes.data, ddst.ptr)))
print("registered host -> registered host %.3f ms" % best(lambda: run(img.ctypes.data, out.ctypes.data)))
print("   identical:", bool(np.array_equal(ref, out)))
