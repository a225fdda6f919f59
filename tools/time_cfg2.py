"""Time cfg2 (4096^2, 5 terms) through the C ABI for a few option settings; DCP_LIB_PATH selects the build."""
import sys
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]
img = np.random.default_rng(c["seed"]).random((H, W), dtype=np.float32)
NR = 20
src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(NR)]
dst = [F.DeviceBuffer(img.nbytes) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])
def t(order, blend, reps=100):
    for w in range(NR):
        F.check(L.dcp_unwarp_image_f32(src[w].ptr, dst[w].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(reps):
        k = r % NR
        F.check(L.dcp_unwarp_image_f32(src[k].ptr, dst[k].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    F.set_option(k, int(v))
    tag += " %s=%s" % (k, v)
for name, order, blend in [("scipy", 1, 0), ("f64lerp", 1, 1), ("f32lerp", 1, 2), ("nearest", 0, 0)]:
    F.debug_counters()
    us = t(order, blend)
    print("%-28s %-8s %.2f us   fallbacks (nofit, vote) per launch: %s" % (tag, name, us, [c / 120.0 for c in F.debug_counters()]), flush=True)
