"""Is a variant build (DCP_LIB_PATH) bit-identical to the oracle on a cfg2-like 1024x1536 frame?  (timing experiments only)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp
from oracle import oracle as orc
orc.build(); orc.set_threads(16)
c = configs.cfg2()
img = np.random.default_rng(3).random((1024, 1536), dtype=np.float32)
a = (img, c["xcenter"] / 3, c["ycenter"] / 3, c["list_fact"])
for blend, ob in (("f64lerp", orc.BLEND_F64LERP), ("scipy", orc.BLEND_SCIPY)):
    got = pp.unwarp_image_backward(*a, blend=blend)
    want = orc.unwarp_image_backward(*a, poly=orc.POLY_KERNEL, blend=ob)
    print(blend, "identical" if np.array_equal(got, want) else "DIFFERENT: %d pixels" % np.count_nonzero(got != want))
