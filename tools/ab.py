"""Interleaved A/B timing of option sets on cfg2 in ONE process (round-robin, median of rounds).
   python tools/ab.py "lds_gather=0" "lds_gather=1" "lds_gather=0,pipe_depth=4" ...   [env DCP_BLEND=1]"""
import os
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
blend = int(os.environ.get("DCP_BLEND", "1")); order = int(os.environ.get("DCP_ORDER", "1"))
DEFAULTS = {k: F.get_option(k) for k in ("tile_rows", "pipe_depth", "xcd_remap", "coef_lds", "lds_gather", "tile_cert", "wg_box", "wg_per_cu")}


def run(reps=60):
    for w in range(NR):
        F.check(L.dcp_unwarp_image_f32(src[w].ptr, dst[w].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(reps):
        k = r % NR
        F.check(L.dcp_unwarp_image_f32(src[k].ptr, dst[k].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / reps * 1e3


sets = sys.argv[1:] or ["lds_gather=0", "lds_gather=1"]
res = {s: [] for s in sets}
fb = {}
for rnd in range(7):
    for s in sets:
        for k, v in DEFAULTS.items():
            F.set_option(k, v)
        for kv in s.split(","):
            if kv:
                k, v = kv.split("="); F.set_option(k, int(v))
        F.debug_counters()
        res[s].append(run())
        fb[s] = [x / 80.0 for x in F.debug_counters()]
for s in sets:
    v = sorted(res[s])
    print("%-44s median %.2f us  min %.2f  max %.2f   fallback tiles/launch (nofit, vote) %s" % (s, v[len(v) // 2], v[0], v[-1], fb[s]), flush=True)
