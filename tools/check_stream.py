"""remap_stream_kernel (wg_box=2) against the oracle on frames with whole and partial tiles (experiment gate)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs, _ffi as F
from discorpy_amd.post import postprocessing as pp
from oracle import oracle as orc
orc.build(); orc.set_threads(32)
F.set_option("wg_box", 2)
c = configs.cfg2()
bad = 0
for (h, w) in ((4096, 4096), (4000, 4100), (3001, 5003), (2560, 2560)):
    img = np.random.default_rng(h).random((h, w), dtype=np.float32)
    s = min(h, w) / 4096.0
    fact = [c["list_fact"][i] / s ** i for i in range(len(c["list_fact"]))]
    a = (img, c["xcenter"] * w / 4096, c["ycenter"] * h / 4096, fact)
    for blend, ob in (("f64lerp", orc.BLEND_F64LERP), ("scipy", orc.BLEND_SCIPY)):
        got = pp.unwarp_image_backward(*a, blend=blend)
        k = F.last_kernel()
        want = orc.unwarp_image_backward(*a, poly=orc.POLY_KERNEL, blend=ob)
        same = np.array_equal(got, want)
        bad += not same
        print(h, w, blend, k, "identical" if same else "DIFFERENT: %d pixels" % np.count_nonzero(got != want), flush=True)
sys.exit(1 if bad else 0)
