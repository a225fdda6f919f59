"""How often does the default blend (float64 factorised lerp) differ from scipy's operation order, and by how much?
GPU kernels, cfg2 map, noise frames in several value ranges."""
import sys
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp
c = configs.cfg2()
tot = diff = 0
worst = 0
for k, (lo, hi) in enumerate([(0, 1), (0, 1), (0, 255), (0, 65535), (-1, 1), (-700, 1300), (0.5, 1.0), (1e-3, 1e3)] * 2):
    img = (np.random.default_rng(100 + k).random(c["shape"], dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    a = pp.unwarp_image_backward(img, c["xcenter"] + k, c["ycenter"] - k, c["list_fact"], blend="scipy")
    b = pp.unwarp_image_backward(img, c["xcenter"] + k, c["ycenter"] - k, c["list_fact"], blend="f64lerp")
    ne = a != b
    n = int(np.count_nonzero(ne))
    if n:
        ia, ib = a[ne].view(np.int32).astype(np.int64), b[ne].view(np.int32).astype(np.int64)
        worst = max(worst, int(np.abs(ia - ib).max()))
    tot += a.size; diff += n
    print("range [%g, %g): %d of %d pixels differ" % (lo, hi, n, a.size), flush=True)
print("total: %d of %d pixels differ (%.2e), largest difference %d float32 ulp" % (diff, tot, diff / tot, worst))
