#!/usr/bin/env python
"""Phase timeline of spline_tile_filter_kernel (the lab build: `make -C discorpy_amd/csrc lab`, DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_lab.so):
per tile, cycles between the phase boundaries [commit+barrier | issue next loads | recursion | store issue | last barrier]."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F, configs  # noqa: E402

L = F.lib()
F.require_device()
c = configs.cfg2()
H, W = c["shape"]
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
src = F.DeviceBuffer(img.nbytes).upload(img)
dst = F.DeviceBuffer(img.nbytes)
fa, n = F.fact_array(c["list_fact"])
order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(3):
    F.check(L.dcp_unwarp_image_spline_f32(src.ptr, dst.ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 0, 1, -1, None))
F.check(L.dcp_stream_synchronize(-1, None))
buf = np.zeros((2, 4096, 8), np.uint64)
L.dcp_experiment_read_tf_trace.argtypes = [C.c_void_p]
assert L.dcp_experiment_read_tf_trace(buf.ctypes.data) == 0
names = ["commit+barrier", "issue next loads", "recursion", "store issue", "last barrier"]
for axis in (0, 1):
    t = buf[axis].astype(np.int64)
    used = t[:, 0] > 0
    t = t[used]
    d = np.diff(t[:, :6], axis=1)
    d[:, 4] = np.where(t[:, 5] > 0, d[:, 4], 0)
    span = (t[:, 4].max() - t[:, 0].min())
    print("axis %d: %d tiles, kernel span %.1f us (at 100 MHz s_memtime ticks: %d ticks)" % (axis, len(t), span / 100.0, span))
    for k, nm in enumerate(names):
        print("   %-18s mean %8.0f ticks  median %8.0f  max %8.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), d[:, k].max()))
    print("   per tile total     mean %8.0f ticks" % (t[:, 4] - t[:, 0]).mean())
    print("   inside the recursion (wave 0): to the end of its causal warm-up %8.0f, first barrier wait %8.0f, rest %8.0f" % (
        (t[:, 6] - t[:, 2]).mean(), (t[:, 7] - t[:, 6]).mean(), (t[:, 3] - t[:, 7]).mean()))

rbuf = np.zeros((2, 4096, 8), np.uint64)
L.dcp_experiment_read_tf_trace_r.argtypes = [C.c_void_p]
assert L.dcp_experiment_read_tf_trace_r(rbuf.ctypes.data) == 0
stages = ["causal warm-up", "barrier", "causal segment", "barrier", "anti-causal warm-up", "barrier", "anti-causal segment"]
for axis in (0, 1):
    t = rbuf[axis].astype(np.int64)
    t = t[t[:, 0] > 0]
    d = np.diff(t, axis=1)
    print("axis %d, the recursion as wave 8 sees it (cycles): " % axis + "  ".join("%s %d" % (nm, d[:, k].mean()) for k, nm in enumerate(stages)))
