"""Quick GPU-side parity + timing probe through the raw C ABI (no torch): run with gpurun."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from discorpy_amd import _ffi as F  # noqa: E402
from oracle import oracle as orc  # noqa: E402

L = F.lib()
print("devices", F.require_device())
orc.set_threads(min(32, orc.max_threads()))

COEF05 = [1.00227490554, -2.99523692178e-05, 8.99519088e-08, -1.57066461911e-10, 8.08880211618e-14]
NAMES = {0: "scipy", 1: "f64lerp", 2: "f32lerp"}


def run_image(img, xc, yc, fact, order, blend, round32=1):
    H, W = img.shape
    src = F.DeviceBuffer(img.nbytes).upload(img)
    dst = F.DeviceBuffer(img.nbytes)
    fa, n = F.fact_array(fact)
    F.check(L.dcp_unwarp_image_f32(src.ptr, dst.ptr, H, W, W, 1, xc, yc, fa, n, order, round32, blend,
                                   F.MEM_DEVICE, -1, None))
    F.check(L.dcp_stream_synchronize(-1, None))
    return dst.download((H, W), np.float32)


def compare(tag, got, ref):
    bad = int((got != ref).sum())
    md = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())
    print("%-40s mismatches %8d / %d  max|d| %.3e" % (tag, bad, got.size, md), flush=True)
    return bad


rng = np.random.default_rng(7)
ok = True
for (H, W, xc, yc, fact) in [(64, 64, 32.0, 32.0, [1.0, 3e-3]),
                             (300, 517, 250.3, 140.9, COEF05),
                             (800, 1280, 588.692801577, 462.092631791, COEF05),
                             (257, 255, 100.0, 128.0, [1.3, 2e-3]),
                             (2, 2, 0.5, 0.5, [1.0]),
                             (1, 7, 3.0, 0.0, [1.0, 1e-2]),
                             (9, 1, 0.0, 4.0, [1.0, 1e-2])]:
    img = (rng.random((H, W), dtype=np.float32) * 255).astype(np.float32)
    for order, blend in [(0, 0), (1, 0), (1, 1), (1, 2)]:
        ref = orc.unwarp_image_backward(img, xc, yc, fact, order=order, poly=orc.POLY_KERNEL,
                                        blend=blend)
        got = run_image(img, xc, yc, fact, order, blend)
        bad = compare("%dx%d n=%d order=%d blend=%s" % (H, W, len(fact), order, NAMES[blend]), got, ref)
        ok &= bad == 0
print("PARITY", "OK" if ok else "FAILED")

# ---- timing: cfg2, ring of frames > 2 GB to defeat the 256 MiB infinity cache
H = W = 4096
s = 1280 / 4096
fact = [a * s ** i for i, a in enumerate(COEF05)]
xc, yc = 588.692801577 / s, 462.092631791 / s
img = np.random.default_rng(20260928).random((H, W), dtype=np.float32)
NR = 20
ring_src = [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(NR)]
ring_dst = [F.DeviceBuffer(img.nbytes) for _ in range(NR)]
fa, n = F.fact_array(fact)
def time_cfg(order, blend, reps=100):
    e0, e1 = F.Event(), F.Event()
    for w in range(NR):
        F.check(L.dcp_unwarp_image_f32(ring_src[w].ptr, ring_dst[w].ptr, H, W, W, 1, xc, yc, fa, n,
                                       order, 1, blend, F.MEM_DEVICE, -1, None))
    e0.record()
    for r in range(reps):
        k = r % NR
        F.check(L.dcp_unwarp_image_f32(ring_src[k].ptr, ring_dst[k].ptr, H, W, W, 1, xc, yc, fa, n,
                                       order, 1, blend, F.MEM_DEVICE, -1, None))
    e1.record()
    e1.synchronize()
    return e0.elapsed_ms(e1) / reps


for pd in (1, 2, 4):
    F.set_option("pipe_depth", pd)
    for tr in (8, 16, 32):
        F.set_option("tile_rows", tr)
        for xr in (0, 1):
            F.set_option("xcd_remap", xr)
            ms = time_cfg(1, 1)
            print("cfg2 f64lerp pd=%d tile_rows=%2d xcd=%d : %.2f us  %.0f Mpix/s  %.2f TB/s algorithmic"
                  % (pd, tr, xr, ms * 1e3, H * W / ms / 1e3, 8.0 * H * W / ms / 1e9), flush=True)
F.set_option("tile_rows", 16)
F.set_option("xcd_remap", 0)
for pd in (2, 4):
    F.set_option("pipe_depth", pd)
    for order, blend in [(1, 0), (1, 1), (1, 2), (0, 0)]:
        ms = time_cfg(order, blend)
        print("cfg2 pd=%d order=%d blend=%-7s : %.2f us  %.0f Mpix/s  %.2f TB/s algorithmic"
              % (pd, order, NAMES[blend], ms * 1e3, H * W / ms / 1e3, 8.0 * H * W / ms / 1e9), flush=True)
F.set_option("pipe_depth", 2)
t = time.time()
ref = orc.unwarp_image_backward(img, xc, yc, fact, order=1, poly=orc.POLY_KERNEL, blend=1)
print("oracle 4096^2 with %d threads: %.3f s" % (orc.lib().orc_get_threads(), time.time() - t))
got = ring_dst[0].download((H, W), np.float32)
F.check(L.dcp_unwarp_image_f32(ring_src[0].ptr, ring_dst[0].ptr, H, W, W, 1, xc, yc, fa, n, 1, 1, 1,
                               F.MEM_DEVICE, -1, None))
F.check(L.dcp_stream_synchronize(-1, None))
got = ring_dst[0].download((H, W), np.float32)
compare("cfg2 full frame f64lerp", got, ref)
