"""One-pass tile prefilter (option spline_tiled=1) against the chunked passes (=0) and the oracle; times of both."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs, _ffi as F
from discorpy_amd.post import postprocessing as pp
from oracle import oracle as orc
orc.build(); orc.set_threads(32)
c = configs.cfg2()
bad = 0
for shape in ((1024, 4096), (2000, 1500)):
    img = np.random.default_rng(5).random(shape, dtype=np.float32)
    for order, mode in [(3, "reflect"), (3, "mirror"), (2, "reflect"), (5, "mirror"), (4, "nearest"), (5, "reflect"), (3, "grid-constant")]:
        a = (img, c["xcenter"] * shape[1] / 4096, 500.0, c["list_fact"])
        want = orc.unwarp_image_backward(*a, order=order, mode=mode, poly=orc.POLY_KERNEL)
        res = {}
        for t in (1, 0):
            F.set_option("x_spline_tiled", t); F.set_option("x_spline_wg", t)
            res[t] = pp.unwarp_image_backward(*a, order=order, mode=mode)
        d1, d0 = np.count_nonzero(res[1] != want), np.count_nonzero(res[0] != want)
        ok = d1 <= 8 and np.max(np.abs(res[1].astype(np.float64) - want)) < 1e-5
        bad += not ok
        print(shape, order, mode, "tiled != oracle: %d px, chunked != oracle: %d px, tiled != chunked: %d px" % (d1, d0, np.count_nonzero(res[1] != res[0])), "OK" if ok else "BAD", flush=True)
print("bad", bad)
L = F.lib(); H, W = 4096, 4096
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
src = F.DeviceBuffer(img.nbytes).upload(img); dst = F.DeviceBuffer(img.nbytes)
fa, n = F.fact_array(c["list_fact"])
for order in (3, 2, 5):
    for t in (0, 1):
        F.set_option("x_spline_tiled", t); F.set_option("x_spline_wg", t)
        def run():
            F.check(L.dcp_unwarp_image_spline_f32(src.ptr, dst.ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 0, 1, -1, None))
        run(); F.check(L.dcp_stream_synchronize(-1, None))
        e0, e1 = F.Event(), F.Event(); e0.record()
        for _ in range(10): run()
        e1.record(); e1.synchronize()
        print("order %d tiled=%d: %.3f ms per 4096^2 frame" % (order, t, e0.elapsed_ms(e1) / 10), flush=True)
sys.exit(1 if bad else 0)
