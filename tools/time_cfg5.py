#!/usr/bin/env python
"""A/B on one box: BASELINE config 5 (8192^2, 9-term fisheye model) on 64 x 32 workgroup tiles (option tall_tiles = 1:
remap_wg_color_kernel, one channel, second tile shape) against the per-wave-box kernel (tall_tiles = 0: remap_lds_kernel)."""
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
c5 = configs.cfg5()
H, W = c5["shape"]
fa, nf = F.fact_array(c5["list_fact"])
img = np.random.default_rng(c5["seed"]).random((H, W), dtype=np.float32)
ring = 4
src = [F.DeviceBuffer(img.nbytes, dev).upload(img) for _ in range(ring)]
dst = [F.DeviceBuffer(img.nbytes, dev) for _ in range(ring)]
outs = {}
for blend, order, name in ((F.BLEND_F64LERP, 1, "f64lerp"), (F.BLEND_SCIPY, 1, "scipy"), (F.BLEND_SCIPY, 0, "nearest")):
    for tall in (1, 0, 1, 0):
        F.set_option("x_tall_tiles", tall)

        def run(i):
            F.check(L.dcp_unwarp_image_f32(src[i % ring].ptr, dst[i % ring].ptr, H, W, W, 1, c5["xcenter"], c5["ycenter"], fa, nf, order, 1, blend,
                                           F.MEM_DEVICE, dev, None))
        t = bench.timed_launches(run, 60, dev, settle_ms=300.0)
        run(0)
        outs[tall] = bench.download(dst[0].ptr, (H, W), dev)
        print("%-8s tall_tiles=%d: %8.2f us  %.3f of 8 TB/s  %s" % (name, tall, t, 8.0 * H * W / (t * 1e-6) / 8e12, F.last_kernel()), flush=True)
    print("   identical: %s" % bool(np.array_equal(outs[0], outs[1])), flush=True)
F.set_option("x_tall_tiles", 1)
