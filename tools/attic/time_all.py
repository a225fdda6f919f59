"""Kernel-only timings (HIP events, device-resident, ring > 256 MiB where it fits) of every BASELINE
configuration through the C ABI.  python tools/time_all.py [key=value options...]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
for kv in sys.argv[1:]:
    k, v = kv.split("="); F.set_option(k, int(v))


def timed(fn, reps):
    for w in range(max(3, min(reps, 60))):      # warm-up long enough for the clocks to settle
        fn(w)
    e0, e1 = F.Event(), F.Event(); e0.record()
    for r in range(reps):
        fn(r)
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


def report(name, us, pixels, bpp):
    print("%-58s %9.2f us %9.0f Mpix/s %6.2f TB/s algorithmic (%.0f%% of 8 TB/s)"
          % (name, us, pixels / us, bpp * pixels / us / 1e6, bpp * pixels / us / 1e6 / 8 * 100), flush=True)


def frames(cfg, nring):
    H, W = cfg["shape"]
    img = np.random.default_rng(cfg["seed"]).random((H, W), dtype=np.float32)
    return H, W, [F.DeviceBuffer(img.nbytes).upload(img) for _ in range(nring)], [F.DeviceBuffer(img.nbytes) for _ in range(nring)]


# ---- cfg2 / cfg3
c = configs.cfg3(); H, W, src, dst = frames(c, 16)
fa, n = F.fact_array(c["list_fact"]); ca, _ = F.fact_array(c["list_coef"])
for name, order, blend in [("scipy", 1, 0), ("f64lerp", 1, 1), ("f32lerp", 1, 2), ("nearest", 0, 0)]:
    us = timed(lambda r: F.check(L.dcp_unwarp_image_f32(src[r % 16].ptr, dst[r % 16].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None)), 80)
    report("cfg2 radial 4096^2 5-term %s" % name, us, H * W, 8)
for name, order, blend in [("f64lerp", 1, 1), ("nearest", 0, 0)]:
    us = timed(lambda r: F.check(L.dcp_perspective_image_f32(src[r % 16].ptr, dst[r % 16].ptr, H, W, W, 1, ca, order, blend, 1, -1, None)), 80)
    report("cfg3a perspective 4096^2 %s" % name, us, H * W, 8)
    us = timed(lambda r: F.check(L.dcp_unwarp_fused_f32(src[r % 16].ptr, dst[r % 16].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, ca, order, blend, 1, -1, None)), 80)
    report("cfg3 fused perspective+radial 4096^2 %s" % name, us, H * W, 8)
def twopass(r):
    F.check(L.dcp_unwarp_image_f32(src[r % 16].ptr, dst[r % 16].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, 1, 1, 1, 1, -1, None))
    F.check(L.dcp_perspective_image_f32(dst[r % 16].ptr, src[(r + 1) % 16].ptr, H, W, W, 1, ca, 1, 1, 1, -1, None))
report("cfg3b two-pass (reference semantics) 4096^2 f64lerp", timed(twopass, 40), H * W, 16)
del src, dst

# ---- cfg5
c = configs.cfg5(); H, W, src, dst = frames(c, 4)
fa, n = F.fact_array(c["list_fact"])
for coef_lds in (0, 1):
    F.set_option("x_coef_lds", coef_lds)
    for name, order, blend in [("f64lerp", 1, 1), ("nearest", 0, 0)]:
        us = timed(lambda r: F.check(L.dcp_unwarp_image_f32(src[r % 4].ptr, dst[r % 4].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None)), 30)
        report("cfg5 radial 8192^2 9-term %s coef_%s" % (name, "LDS" if coef_lds else "SGPR"), us, H * W, 8)
F.set_option("x_coef_lds", 0)
del src, dst

# ---- cfg4: stack kernel on a depth-64 sample of the 2560^2 stack (device-generated data would be the same speed)
c = configs.cfg4(depth=64); D, H, W = c["shape"]
vol = F.DeviceBuffer(D * H * W * 4)
chunk = np.random.default_rng(c["seed"]).random((8, H, W), dtype=np.float32)
for i in range(D // 8):
    F.check(L.dcp_memcpy(vol.ptr + i * chunk.nbytes, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, -1, None))
fa, n = F.fact_array(c["list_fact"])
for nrows, row0, r32, label in [(1, 1277.0, 0, "unwarp_slice_backward (1 row, f64 coords)"), (64, 1000.0, 1, "chunk 64 rows"),
                                (H, 0.0, 1, "all 2560 rows (full corrected stack)")]:
    out = F.DeviceBuffer(D * nrows * W * 4)
    for dch in (8, 16, 64):
        F.set_option("x_d_chunk", dch)
        us = timed(lambda r: F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr, D, H, W, H * W, W, c["xcenter"], c["ycenter"], fa, n, row0, nrows, r32, 1, 1, -1, None)), 10 if nrows > 64 else 50)
        report("cfg4 stack D=%d %s d_chunk=%d" % (D, label, dch), us, D * nrows * W, 12 if nrows == 1 else 8)
    del out
