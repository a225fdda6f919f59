#!/usr/bin/env python
"""Phase timeline of spline_prefilter2d_kernel (the lab build: `make -C discorpy_amd/csrc lab`, DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_lab.so):
per step of wave 0 of every workgroup, s_memtime ticks (100 MHz) between the phase boundaries."""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
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
print(F.last_kernel())
buf = np.zeros((1024, 8, 8), np.uint64)
L.dcp_experiment_read_pf2d_trace.argtypes = [C.c_void_p]
assert L.dcp_experiment_read_pf2d_trace(buf.ctypes.data) == 0
t = buf.astype(np.int64)
used = t[:, 0, 0] > 0
t = t[used]
print("%d workgroups traced; kernel span %.1f us" % (len(t), (t[t > 0].max() - t[:, 0, 0].min()) / 100.0))
names = ["column anti-causal + LDS writes + prefetch issue", "barrier 1 (tile complete)", "row recursions (91 LDS reads, 148 steps)", "barrier 2 + write-back + barrier 3",
         "store-out (LDS -> global)", "shift + causal extension (waits for the prefetch)", "barrier 4"]
for step in range(8):
    s = t[:, step, :]
    ok = (s[:, 0] > 0) & (s[:, 5] > 0)
    if not ok.any():
        continue
    s = s[ok]
    d = np.diff(s[:, :8], axis=1).astype(np.float64)
    has_next = s[:, 7] > 0
    line = "step %d (%4d workgroups): " % (step, len(s))
    parts = []
    for k, nm in enumerate(names):
        col = d[:, k] if k < 5 else d[has_next, k]
        if len(col):
            parts.append("%s %.0f" % (nm.split(" (")[0], col.mean()))
    print(line + " | ".join(parts) + " | step total %.0f ticks" % ((s[has_next, 7] - s[has_next, 0]).mean() if has_next.any() else (s[:, 5] - s[:, 0]).mean()))
first = t[:, 0, 0] - t[:, 0, 0].min()
print("start of step 0 after the first workgroup's: mean %.0f ticks, max %.0f (the initial 34 + 66 row loads and their recursions come before it)" % (first.mean(), first.max()))
