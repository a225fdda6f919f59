#!/usr/bin/env python
"""Per-launch time of the frame kernel against the size of the ring of frames it cycles through (24 = the bench's: 3 GB, every
launch streams from and to HBM; 4 and 1: the working set stays in the 256 MB memory-side cache), as a timeline of 40 blocks of 100
launches -- f32 blend and the default f64lerp.  Shows which part of a launch is the memory system's and that the f32 blend's
slow episodes are HBM-side.   python tools/ring_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device(); dev = -1
c2 = configs.cfg2(); H, W = c2["shape"]
fa, nf = F.fact_array(c2["list_fact"])
f32 = np.random.default_rng(3).random((H, W), dtype=np.float32)
nmax = 24
src = [F.DeviceBuffer(f32.nbytes, dev).upload(f32) for _ in range(nmax)]
dst = [F.DeviceBuffer(f32.nbytes, dev) for _ in range(nmax)]
for name, blend in (("f32lerp", F.BLEND_F32LERP), ("f64lerp", F.BLEND_F64LERP)):
    for n in (24, 4, 1):
        def run(i):
            F.check(L.dcp_unwarp_image_f32(src[i % n].ptr, dst[i % n].ptr, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf, 1, 1, blend, F.MEM_DEVICE, dev, None))
        blocks, per = 40, 100
        ev = [F.Event(dev) for _ in range(blocks + 1)]
        for i in range(200): run(i)
        F.check(L.dcp_stream_synchronize(dev, None))
        k = 0
        for b in range(blocks):
            ev[b].record()
            for _ in range(per):
                run(k); k += 1
        ev[blocks].record(); ev[blocks].synchronize()
        us = [ev[b].elapsed_ms(ev[b + 1]) * 1e3 / per for b in range(blocks)]
        print(name, "ring", n, " ".join("%.1f" % u for u in us), flush=True)
