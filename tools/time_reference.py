"""Time the REFERENCE itself (discorpy 1.7.0 from /root/reference, numpy + scipy, one core) on the BASELINE
configurations -- in the build container only (the reference does not travel to the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import discorpy.post.postprocessing as post  # noqa: E402
from discorpy_amd import configs  # noqa: E402


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)


print("host: %d cores visible, numpy %s" % (os.cpu_count(), np.__version__))
img = np.random.default_rng(2).random(configs.DOT_05_SHAPE, dtype=np.float32) * 255
t = best(lambda: post.unwarp_image_backward(img, configs.XCENTER_DOT_05, configs.YCENTER_DOT_05, configs.COEF_DOT_05), 5)
print("cfg1 800x1280 unwarp_image_backward:            %8.1f ms  %7.1f Mpix/s" % (t * 1e3, img.size / t / 1e6))
c = configs.cfg2(); img = np.random.default_rng(c["seed"]).random(c["shape"], dtype=np.float32)
t = best(lambda: post.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"]))
print("cfg2 4096^2 unwarp_image_backward order 1:      %8.1f ms  %7.1f Mpix/s" % (t * 1e3, img.size / t / 1e6))
t0 = best(lambda: post.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], order=0), 2)
print("cfg2 4096^2 unwarp_image_backward order 0:      %8.1f ms  %7.1f Mpix/s" % (t0 * 1e3, img.size / t0 / 1e6))
c3 = configs.cfg3()
tp = best(lambda: post.correct_perspective_image(img, c3["list_coef"]), 2)
print("cfg3 4096^2 correct_perspective_image:          %8.1f ms  %7.1f Mpix/s" % (tp * 1e3, img.size / tp / 1e6))
print("cfg3 two-pass (radial then perspective):        %8.1f ms" % ((t + tp) * 1e3))
t3 = best(lambda: post.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], order=3), 1)
print("cfg2 4096^2 unwarp_image_backward order 3:      %8.1f ms  %7.1f Mpix/s" % (t3 * 1e3, img.size / t3 / 1e6))
c4 = configs.cfg4(64)
vol = np.random.default_rng(4).random((64, 2560, 2560), dtype=np.float32)
ts = best(lambda: post.unwarp_slice_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1277))
print("cfg4 depth 64 unwarp_slice_backward:            %8.1f ms" % (ts * 1e3))
tc = best(lambda: post.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1000, 1063), 2)
print("cfg4 depth 64 unwarp_chunk_slices 64 rows:      %8.1f ms  %7.1f Mvoxel/s" % (tc * 1e3, 64 * 64 * 2560 / tc / 1e6))
