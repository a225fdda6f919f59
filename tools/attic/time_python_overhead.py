#!/usr/bin/env python
"""What a caller of the Python drop-in sees per call on DEVICE-resident frames (torch tensors): wall time per call of
post.unwarp_image_backward in a loop (results dropped / written into out=), against the bare C-ABI call and the kernel time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402

F.require_device()
L = F.lib()
for shape in ((4096, 4096), (1024, 1024), (256, 256)):
    c = configs.cfg2()
    s = 4096.0 / shape[1]
    xc, yc = c["xcenter"] / s, c["ycenter"] / s
    fact = [v * s ** k for k, v in enumerate(c["list_fact"])]
    ring = [torch.rand(shape, device="cuda") for _ in range(8)]
    outs = [torch.empty(shape, device="cuda") for _ in range(8)]
    fa, nf = F.fact_array(fact)
    H, W = shape
    st = torch.cuda.current_stream().cuda_stream

    def wall(fn, n=400):
        for i in range(20):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6 / n, t_enq * 1e6 / n
    a = wall(lambda i: pp.unwarp_image_backward(ring[i % 8], xc, yc, fact))
    b = wall(lambda i: pp.unwarp_image_backward(ring[i % 8], xc, yc, fact, out=outs[i % 8]))
    d = wall(lambda i: L.dcp_unwarp_image_f32(ring[i % 8].data_ptr(), outs[i % 8].data_ptr(), H, W, W, 1, xc, yc, fa, nf, 1, 1, F.BLEND_F64LERP,
                                              F.MEM_DEVICE, -1, st))
    print("%dx%d: pp.unwarp_image_backward %.1f us per call (host side %.1f); with out= %.1f (%.1f); bare C ABI %.1f (%.1f)  [%s]" % (
        H, W, a[0], a[1], b[0], b[1], d[0], d[1], F.last_kernel()), flush=True)
