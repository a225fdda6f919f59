import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from discorpy_amd import _ffi as F
from discorpy_amd import configs
L = F.lib(); F.require_device(); dev = -1
c5 = configs.cfg5(); H, W = c5["shape"]
fa, nf = F.fact_array(c5["list_fact"])
img = np.random.default_rng(c5["seed"]).random((H, W), dtype=np.float32)
src = F.DeviceBuffer(img.nbytes, dev).upload(img); dst = F.DeviceBuffer(img.nbytes, dev)
outs = {}
for tall in (1, 0):
    F.set_option("tall_tiles", tall)
    F.check(L.dcp_unwarp_image_f32(src.ptr, dst.ptr, H, W, W, 1, c5["xcenter"], c5["ycenter"], fa, nf, 1, 1, F.BLEND_SCIPY, F.MEM_DEVICE, dev, None))
    outs[tall] = bench.download(dst.ptr, (H, W), dev)
    print(F.last_kernel())
d = np.argwhere(outs[0] != outs[1])
print("differing", len(d))
if len(d):
    ys, xs = d[:, 0], d[:, 1]
    print("y range", ys.min(), ys.max(), "x range", xs.min(), xs.max())
    ty, tx = ys // 32, xs // 64
    tiles = np.unique(np.stack([ty, tx], 1), axis=0)
    print("tiles affected", len(tiles), "of", (H // 32) * (W // 64), "first", tiles[:10].tolist())
    for (y, x) in d[:8]:
        print(y, x, outs[1][y, x], outs[0][y, x], "row in tile", y % 32, "col in tile", x % 64)
    print("rows-in-tile histogram", np.bincount(ys % 32, minlength=32).tolist())
    print("cols-in-tile histogram (16 bins)", np.bincount((xs % 64) // 4, minlength=16).tolist())
