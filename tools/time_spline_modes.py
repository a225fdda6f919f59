import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]; fa, nf = F.fact_array(c["list_fact"])
rng = np.random.default_rng(2)
srcs = [F.DeviceBuffer(H*W*4, -1).upload(rng.random((H, W), dtype=np.float32)) for _ in range(4)]
dsts = [F.DeviceBuffer(H*W*4, -1) for _ in range(4)]
MODES = {"reflect": 0, "nearest": None}
# boundary mode indices as the Python front end passes them
from discorpy_amd.post import postprocessing as pp
for order in (2, 3, 4, 5):
    for mode in ("reflect", "mirror", "nearest", "grid-constant", "constant", "wrap"):
        m = pp._spline_mode(mode, None)
        def run(i):
            F.check(L.dcp_unwarp_image_spline_f32(srcs[i % 4].ptr, dsts[i % 4].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, order, m, F.MEM_DEVICE, -1, None))
        t = bench.timed_launches(run, 12, -1, settle_ms=150.0)
        print("order %d %-14s %8.1f us  %s" % (order, mode, t, F.last_kernel()), flush=True)
