#!/usr/bin/env python
"""Same-box A/B of stack_wg_kernel's tile order: option xcd_remap = 1 (each XCD a contiguous run of neighbouring tiles, depth chunk after
depth chunk), 2 (the default: those runs for integer element types only) and 0 (grid order: neighbours on different XCDs), float32 and uint16 shards of config 4; identical outputs required.
us per launch (HIP events after 100 ms of the same launches).   python tools/ab_stack_order.py [depth]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402

L = F.lib()
F.require_device()
dev = -1
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c4 = configs.cfg4(D)
D, Hs, Ws = c4["shape"]
f4, n4 = F.fact_array(c4["list_fact"])
rng = np.random.default_rng(3)
for name, dt, scale in (("float32", np.float32, 1.0), ("uint16", np.uint16, 60000.0)):
    es = np.dtype(dt).itemsize
    vol = F.DeviceBuffer(D * Hs * Ws * es, dev)
    out = F.DeviceBuffer(D * Hs * Ws * es, dev)
    chunk = (rng.random((4, Hs, Ws)) * scale).astype(dt)
    for d in range(0, D, 4):
        F.check(L.dcp_memcpy(vol.ptr + d * Hs * Ws * es, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, dev, None))

    def shard(_i):
        if dt is np.float32:
            F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr, D, Hs, Ws, Hs * Ws, Ws, c4["xcenter"], c4["ycenter"], f4, n4, 0.0, Hs, 1, F.BLEND_F64LERP,
                                                F.MEM_DEVICE, dev, None))
        else:
            F.check(L.dcp_unwarp_stack_rows_typed(vol.ptr, out.ptr, F.DTYPE_BY_NAME[name], 0, D, Hs, Ws, Hs * Ws, Ws, c4["xcenter"], c4["ycenter"],
                                                  f4, n4, 0.0, Hs, 1, F.MEM_DEVICE, dev, None))
    opt = os.environ.get("AB_OPTION", "x_xcd_remap")       # AB_OPTION=x_store_wait AB_MODES=1,0: another option of the stack kernel
    modes = [int(m) for m in os.environ.get("AB_MODES", "2,1,0").split(",")]       # (one mode, few launches: the counter passes)
    outs, ts = {m: None for m in modes}, {m: [] for m in modes}
    for rep in range(int(os.environ.get("AB_REPS", "3"))):
        for mode in modes:
            F.set_option(opt, mode)
            ts[mode].append(bench.timed_launches(shard, int(os.environ.get("AB_LAUNCHES", "12")), dev, settle_ms=float(os.environ.get("AB_SETTLE_MS", "100"))))
            g = np.empty((2, Hs, Ws), dt)
            F.check(L.dcp_memcpy(g.ctypes.data, out.ptr + (D - 3) * Hs * Ws * es, g.nbytes, F.COPY_D2H, dev, None))
            outs[mode] = g
    F.set_option(opt, int(os.environ.get("AB_DEFAULT", "2")))
    alg = 2.0 * D * Hs * Ws * es
    got = list(outs.values())
    print("%-7s shard (%d, %d, %d), option %s: %s   identical %s   (%s)" % (
        name, D, Hs, Ws, opt, "   ".join("%d: %s us (%.3f of 8 TB/s)" % (m, ["%.1f" % t for t in ts[m]], alg / (min(ts[m]) * 1e-6) / 8e12) for m in modes),
        all(np.array_equal(got[0], o) for o in got[1:]), F.last_kernel()), flush=True)
    vol.free()
    out.free()
