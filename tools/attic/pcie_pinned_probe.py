#!/usr/bin/env python
"""Does the HIP runtime in use (PyTorch's bundled one by default, /opt/rocm's with DISCORPY_AMD_SYSTEM_HIP=1) move data in both PCIe
directions at once when the host buffers are REGISTERED (hipHostRegister) and the copies asynchronous on two streams?
64 MiB up, 64 MiB down: one after the other, together; pageable for comparison."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F  # noqa: E402

F.lib()
F.require_device()
path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
hip = C.CDLL(path)
print("runtime:", path)
n = 64 << 20
up = np.ones(n, np.uint8)
down = np.zeros(n, np.uint8)
d0, d1 = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(d0), C.c_size_t(n)) == 0 and hip.hipMalloc(C.byref(d1), C.c_size_t(n)) == 0
s0, s1 = C.c_void_p(), C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s0), 1) == 0 and hip.hipStreamCreateWithFlags(C.byref(s1), 1) == 0
H2D, D2H = 1, 2


def t(fn, reps=7):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def both_async():
    hip.hipMemcpyAsync(d0, C.c_void_p(up.ctypes.data), C.c_size_t(n), H2D, s0)
    hip.hipMemcpyAsync(C.c_void_p(down.ctypes.data), d1, C.c_size_t(n), D2H, s1)
    hip.hipStreamSynchronize(s0)
    hip.hipStreamSynchronize(s1)


def serial():
    hip.hipMemcpyAsync(d0, C.c_void_p(up.ctypes.data), C.c_size_t(n), H2D, s0)
    hip.hipStreamSynchronize(s0)
    hip.hipMemcpyAsync(C.c_void_p(down.ctypes.data), d1, C.c_size_t(n), D2H, s1)
    hip.hipStreamSynchronize(s1)

print("pageable: one after the other %.3f ms, both issued then waited %.3f ms" % (t(serial), t(both_async)))
t0 = time.perf_counter()
r0 = hip.hipHostRegister(C.c_void_p(up.ctypes.data), C.c_size_t(n), 0)
r1 = hip.hipHostRegister(C.c_void_p(down.ctypes.data), C.c_size_t(n), 0)
print("hipHostRegister 2 x 64 MiB: rc %d %d, %.3f ms" % (r0, r1, (time.perf_counter() - t0) * 1e3))
print("registered: one after the other %.3f ms, both issued then waited %.3f ms" % (t(serial), t(both_async)))
t0 = time.perf_counter()
hip.hipHostUnregister(C.c_void_p(up.ctypes.data))
hip.hipHostUnregister(C.c_void_p(down.ctypes.data))
print("hipHostUnregister: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
