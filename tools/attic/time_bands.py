import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
from discorpy_amd.post import postprocessing as pp
c = configs.cfg2(); img = np.random.default_rng(1).random(c["shape"], dtype=np.float32)
out = np.zeros_like(img)
def best(fn, n=9):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
for nb in (2, 3, 4, 6, 8, 12, 16, 24):
    F.set_option("host_bands", nb)
    print("bands %2d: %.2f ms" % (nb, best(lambda: pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], out=out))), flush=True)
