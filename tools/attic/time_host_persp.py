import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp
c = configs.cfg3(); img = np.random.default_rng(1).random(c["shape"], dtype=np.float32); out = np.zeros_like(img)
def best(fn, n=9):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
print("cfg3a perspective 4096^2 numpy->numpy: %.2f ms" % best(lambda: pp.correct_perspective_image(img, c["list_coef"], out=out)))
print("cfg2 radial 4096^2 numpy->numpy:       %.2f ms" % best(lambda: pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], out=out)))
