import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]; fa, nf = F.fact_array(c["list_fact"])
rng = np.random.default_rng(2)
srcs = [F.DeviceBuffer(H*W*4, -1).upload(rng.random((H, W), dtype=np.float32)) for _ in range(8)]
dsts = [F.DeviceBuffer(H*W*4, -1) for _ in range(8)]
outs = {}
for rep in range(3):
    for xcd in (0, 1):
        F.set_option("x_spline_xcd", xcd)
        for order in (3, 2):
            def run(i):
                F.check(L.dcp_unwarp_image_spline_f32(srcs[i % 8].ptr, dsts[i % 8].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, order, 0, F.MEM_DEVICE, -1, None))
            t = bench.timed_launches(run, 30, -1, settle_ms=300.0)
            run(0)
            outs[(xcd, order)] = bench.download(dsts[0].ptr, (H, W), -1)
            print("order %d xcd order %d: %8.2f us" % (order, xcd, t), flush=True)
for order in (3, 2):
    print("order", order, "pixels differing between tile orders:", int(np.count_nonzero(outs[(0, order)] != outs[(1, order)])))
