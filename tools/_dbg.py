import sys
sys.path.insert(0, '.')
import numpy as np
from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp
from oracle import oracle as orc
from tests.conftest import noise
shape=(63,257)
img = (noise(hash(shape) % 1000, shape) * 255).astype(np.float32)
xc, yc = 0.43 * shape[1], 0.61 * shape[0]
fact = [1.01, -4e-4, 3e-7]
for blend, ob in (("scipy", 0), ("f64lerp", 1)):
    want = orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL, blend=ob)
    for lg in (0, 1):
        F.set_option("lds_gather", lg)
        F.debug_counters()
        got = pp.unwarp_image_backward(img, xc, yc, fact, blend=blend)
        bad = np.argwhere(got != want)
        print(blend, "lds", lg, "mismatches", len(bad), "counters", F.debug_counters())
        for (y, x) in bad[:12]:
            print("   ", y, x, got[y, x], want[y, x])
        if len(bad):
            print("   rows", sorted(set(bad[:,0]))[:20], "cols", sorted(set(bad[:,1]))[:20])
