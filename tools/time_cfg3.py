#!/usr/bin/env python
"""A/B on one box: BASELINE config 3 (4096^2, 3 x 3 homography fused with the 5-term radial model) on remap_wg_kernel -- one source
box per 128 x 32 workgroup tile, taken from the corners of the tile's perspective bounding box under the host's certificate
(option fused_wg = 1, round 5) -- against the per-wave-box kernel with the per-pixel containment vote (fused_wg = 0, rounds 1-4)."""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
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
c3 = configs.cfg3()
H, W = c3["shape"]
fa, nf = F.fact_array(c3["list_fact"])
ca, _ = F.fact_array(c3["list_coef"])
ring = 16
rng = np.random.default_rng(c3["seed"])
src = [F.DeviceBuffer(H * W * 4, dev).upload(rng.random((H, W), dtype=np.float32)) for _ in range(ring)]
dst = [F.DeviceBuffer(H * W * 4, dev) for _ in range(ring)]
outs = {}
for blend, order, name in ((F.BLEND_F64LERP, 1, "f64lerp"), (F.BLEND_SCIPY, 1, "scipy"), (F.BLEND_SCIPY, 0, "nearest")):
    for wg in (1, 0, 1, 0, 1, 0):
        F.set_option("x_fused_wg", wg)

        def run(i):
            F.check(L.dcp_unwarp_fused_f32(src[i % ring].ptr, dst[i % ring].ptr, H, W, W, 1, c3["xcenter"], c3["ycenter"], fa, nf, ca, order, blend,
                                           F.MEM_DEVICE, dev, None))
        t = bench.timed_launches(run, 200, dev, settle_ms=300.0)
        run(0)
        outs[wg] = bench.download(dst[0].ptr, (H, W), dev)
        print("%-8s fused_wg=%d: %8.2f us  %.3f of 8 TB/s  %s" % (name, wg, t, 8.0 * H * W / (t * 1e-6) / 8e12, F.last_kernel()), flush=True)
    print("   identical: %s" % bool(np.array_equal(outs[0], outs[1])), flush=True)
F.set_option("x_fused_wg", 1)
cnt = F.debug_counters() if hasattr(F, "debug_counters") else None
print("counters (tiles whose box did not fit, ...):", cnt)
