#!/usr/bin/env python
"""us per 4096^2 launch of the frame kernels by element type / sampler, no verification (for the ablation variants of
tools/variants.sh, selected with DCP_LIB_PATH).  python tools/time_frames.py [label]"""
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

label = sys.argv[1] if len(sys.argv) > 1 else "full"
L = F.lib()
F.require_device()
dev = -1
c2 = configs.cfg2()
H, W = c2["shape"]
fa, nf = F.fact_array(c2["list_fact"])
rng = np.random.default_rng(3)
res = []
f32 = rng.random((H, W), dtype=np.float32)
src = [F.DeviceBuffer(f32.nbytes, dev).upload(f32) for _ in range(12)]
dst = [F.DeviceBuffer(f32.nbytes, dev) for _ in range(12)]
for name, order, blend in (("f32/f64lerp", 1, F.BLEND_F64LERP), ("f32/scipy", 1, F.BLEND_SCIPY), ("f32/f32lerp", 1, F.BLEND_F32LERP), ("f32/nearest", 0, F.BLEND_SCIPY)):
    def run(i):
        F.check(L.dcp_unwarp_image_f32(src[i % 12].ptr, dst[i % 12].ptr, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf, order, 1, blend, F.MEM_DEVICE, dev, None))
    res.append("%s %.2f" % (name, bench.timed_launches(run, 600, dev, settle_ms=250.0)))
for b in src + dst:
    b.free()
for name, dt, scale in (("uint16", np.uint16, 60000.0), ("uint8", np.uint8, 255.0)):
    img = (rng.random((H, W)) * scale).astype(dt)
    code = F.DTYPE_BY_NAME[name]
    src = [F.DeviceBuffer(img.nbytes, dev).upload(img) for _ in range(12)]
    dst = [F.DeviceBuffer(img.nbytes, dev) for _ in range(12)]
    for order in (1, 0):
        def run(i):
            F.check(L.dcp_unwarp_image_typed(src[i % 12].ptr, dst[i % 12].ptr, code, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf, order, 0, F.MEM_DEVICE, dev, None))
        res.append("%s/order%d %.2f" % (name, order, bench.timed_launches(run, 600, dev, settle_ms=250.0)))
    for b in src + dst:
        b.free()
print("%-8s " % label + "   ".join(res), flush=True)
