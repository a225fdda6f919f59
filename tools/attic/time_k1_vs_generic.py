#!/usr/bin/env python
"""A/B on one box: BASELINE config 2 through remap_wg_kernel (128 x 32 tiles, the tuned single-plane kernel) and through the generic
interleaved-pixel kernel with one float32 channel on 128 x 16 tiles (option tall_tiles = 2)."""
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
c = configs.cfg2()
H, W = c["shape"]
fa, nf = F.fact_array(c["list_fact"])
rng = np.random.default_rng(1)
ring = 24
src = [F.DeviceBuffer(H * W * 4, dev).upload(rng.random((H, W), dtype=np.float32)) for _ in range(ring)]
dst = [F.DeviceBuffer(H * W * 4, dev) for _ in range(ring)]
for blend, order, name in ((F.BLEND_F64LERP, 1, "f64lerp"), (F.BLEND_SCIPY, 1, "scipy"), (F.BLEND_SCIPY, 0, "nearest")):
    outs = {}
    for opt in (0, 2, 0, 2):
        F.set_option("x_tall_tiles", opt)

        def run(i):
            F.check(L.dcp_unwarp_image_f32(src[i % ring].ptr, dst[i % ring].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, order, 1, blend, F.MEM_DEVICE, dev, None))
        t = bench.timed_launches(run, 480, dev, settle_ms=300.0)
        run(0)
        outs[opt] = bench.download(dst[0].ptr, (H, W), dev)
        print("%-8s %8.2f us  %.3f  %s" % (name, t, 8.0 * H * W / (t * 1e-6) / 8e12, F.last_kernel()), flush=True)
    print("   identical:", bool(np.array_equal(outs[0], outs[2])), flush=True)
F.set_option("x_tall_tiles", 0)
