"""End-to-end (NumPy in -> NumPy out, PCIe included) timings of the drop-in functions."""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp

def best(fn, n=7):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)

c = configs.cfg2(); img = np.random.default_rng(1).random(c["shape"], dtype=np.float32)
t = best(lambda: pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"]))
print("cfg2 4096^2 numpy->numpy: %.2f ms  %.0f Mpix/s (PCIe-inclusive)" % (t * 1e3, img.size / t / 1e6))
small = np.random.default_rng(2).random(configs.DOT_05_SHAPE, dtype=np.float32) * 255
t = best(lambda: pp.unwarp_image_backward(small, configs.XCENTER_DOT_05, configs.YCENTER_DOT_05, configs.COEF_DOT_05))
print("cfg1-sized 800x1280 numpy->numpy: %.3f ms  %.0f Mpix/s (reference on 1 core: 83 ms)" % (t * 1e3, small.size / t / 1e6))
c5 = configs.cfg5(); img5 = np.random.default_rng(3).random(c5["shape"], dtype=np.float32)
t = best(lambda: pp.unwarp_image_backward(img5, c5["xcenter"], c5["ycenter"], c5["list_fact"]), 3)
print("cfg5 8192^2 numpy->numpy: %.2f ms  %.0f Mpix/s" % (t * 1e3, img5.size / t / 1e6))
vol = np.random.default_rng(4).random((256, 2560, 2560), dtype=np.float32)
c4 = configs.cfg4(256)
t = best(lambda: pp.unwarp_slice_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1277), 5)
print("cfg4 unwarp_slice_backward depth 256 numpy->numpy: %.2f ms (reference: 26.5 ms)" % (t * 1e3))
t = best(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1000, 1063), 3)
print("cfg4 unwarp_chunk_slices_backward 64 rows depth 256 numpy->numpy: %.1f ms (reference: 2590 ms)" % (t * 1e3))
