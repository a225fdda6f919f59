"""Leak check: 3000 randomised calls through the Python front end (NumPy and torch inputs, every entry point); prints the
process RSS and the free device memory every 500 calls -- both must stay flat.    python tools/soak.py"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import sys, time, os, resource
import numpy as np, torch
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, _pool
from discorpy_amd.post import postprocessing as pp
from discorpy_amd.util import utility as util
rng = np.random.default_rng(0)
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
def free_gb(): return torch.cuda.mem_get_info()[0] / 1e9
t0 = time.time(); marks = []
for it in range(3000):
    h, w = int(rng.integers(16, 1200)), int(rng.integers(16, 1200))
    kind = it % 6
    img = rng.random((h, w), dtype=np.float32)
    a = (w * 0.5, h * 0.5, [1.0, 1e-3 / max(h, w), 1e-9])
    if kind == 0: pp.unwarp_image_backward(img, *a)
    elif kind == 1: pp.unwarp_image_backward(torch.from_numpy(img).cuda(), *a)
    elif kind == 2: pp.unwarp_image_backward(img, *a, order=3)
    elif kind == 3: pp.unwarp_chunk_slices_backward(rng.random((3, h, w), dtype=np.float32), *a, 2, min(h - 1, 12))
    elif kind == 4: util.unwarp_color_image_backward((rng.random((h, w, 3)) * 255).astype(np.uint8), *a)
    else: pp.correct_perspective_image(img.astype(np.uint16), [1, 0.01, 1, 0, 1, 2, 1e-5, 0])
    if it % 50 == 7:        # round 3's paths: a host frame large enough for the registered-output path, a batch (stream-ordered scratch), a centre grid
        big = rng.random((4096, 2048 + 64 * (it % 3)), dtype=np.float32)
        pp.unwarp_image_backward(big, big.shape[1] * 0.5, 2048.0, [1.0, 1e-6, 1e-10])
        fr = [torch.from_numpy(rng.random((300, 520), dtype=np.float32)).cuda() for _ in range(5)]
        pp.unwarp_images_backward(fr, [260.0 + i for i in range(5)], [150.0] * 5, [[1.0, 2e-5, 1e-9]] * 5)
        pp.unwarp_slice_backward_centres(rng.random((4, 200, 300), dtype=np.float32), [150.0, 151.0, 152.0], [100.0] * 3, [1.0, 1e-5], 77)
    if it % 500 == 499:
        torch.cuda.synchronize()
        marks.append((it + 1, round(rss()), round(free_gb(), 2), _pool.stats()["idle_bytes"] >> 20))
print("calls, max RSS MB, device free GB, pool idle MB:", marks, "in %.1f s" % (time.time() - t0))
F.release_scratch(); torch.cuda.synchronize(); torch.cuda.empty_cache()
print("after release_scratch: device free GB %.2f" % free_gb())
