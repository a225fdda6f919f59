#!/usr/bin/env python
"""(Lab record: needs the library of commit adad739, where `spline_wg_rowfused_kernel` and the option `x_spline_rowfuse` exist -- the
kernel was measured and NOT kept: profiles/LAB_NOTEBOOK.md, round 5; `tools/variant_from_git.sh` builds from a revision.)
A/B on one box: orders 2 / 3 on a device-resident 4096^2 float32 frame (mode reflect) with the row prefilter folded into the
gather's box staging (option x_spline_rowfuse = 1: spline_col_lds_kernel + spline_wg_rowfused_kernel, round 5) against the three
launches of round 4 (= 0: + spline_row_lds_kernel + spline_wg_kernel).  Pixels differing between the two and against the oracle."""
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
cfg = configs.cfg2()
H, W = cfg["shape"]
fa, nf = F.fact_array(cfg["list_fact"])
rng = np.random.default_rng(2)
ring = 8
imgs = [rng.random((H, W), dtype=np.float32) for _ in range(2)]
srcs = [F.DeviceBuffer(H * W * 4, dev).upload(imgs[i % 2]) for i in range(ring)]
dsts = [F.DeviceBuffer(H * W * 4, dev) for _ in range(ring)]
orc = bench.oracle_module(0)
for order in (3, 2):
    def run(i):
        F.check(L.dcp_unwarp_image_spline_f32(srcs[i % ring].ptr, dsts[i % ring].ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf, order, 0,
                                              F.MEM_DEVICE, dev, None))
    outs = {}
    for fuse in (1, 0, 1, 0):
        F.set_option("x_spline_rowfuse", fuse)
        t = bench.timed_launches(run, 60, dev, settle_ms=300.0)
        run(0)
        outs[fuse] = bench.download(dsts[0].ptr, (H, W), dev)
        print("order %d rowfuse=%d: %8.2f us  %s" % (order, fuse, t, F.last_kernel()), flush=True)
    F.set_option("x_spline_rowfuse", 1)
    d = outs[0] != outs[1]
    print("   pixels differing between the two: %d of %d (max |diff| %.3g)" % (int(d.sum()), d.size, float(np.max(np.abs(outs[0].astype(np.float64) - outs[1]))) if d.any() else 0.0))
    want = orc.unwarp_image_backward(imgs[0], cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], order=order, mode="reflect", poly=orc.POLY_KERNEL)
    for fuse in (1, 0):
        dd = outs[fuse] != want
        print("   rowfuse=%d against the oracle: %d pixels differ, max |diff| %.3g" % (fuse, int(dd.sum()), float(np.max(np.abs(outs[fuse].astype(np.float64) - want)))), flush=True)
print("counters (no fit, vote failed, fused overflow):", F.debug_counters(n=3))
