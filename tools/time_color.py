"""util.unwarp_color_image_backward on a 4096 x 4096 x 3 image: interleaved kernel vs per-channel planes."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.util import utility as util
c = configs.cfg2()


def best(fn, n=5):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


for dt in (np.float32, np.uint8):
    rgb = (np.random.default_rng(1).random((4096, 4096, 3)) * 255).astype(dt)
    a = (rgb, c["xcenter"], c["ycenter"], c["list_fact"])
    print("%-8s numpy -> numpy  interleaved kernel: %7.2f ms   per-channel planes (blend='f32'): %7.2f ms"
          % (np.dtype(dt).name, best(lambda: util.unwarp_color_image_backward(*a)),
             best(lambda: util.unwarp_color_image_backward(*a, blend="f32") if dt == np.float32 else
                  np.stack([__import__("discorpy_amd").post.postprocessing.unwarp_image_backward(np.ascontiguousarray(rgb[:, :, k]), *a[1:]) for k in range(3)], axis=2))), flush=True)
import torch
t = torch.from_numpy((np.random.default_rng(1).random((4096, 4096, 3)) * 255).astype(np.float32)).cuda()


def dev(blend):
    util.unwarp_color_image_backward(t, c["xcenter"], c["ycenter"], c["list_fact"], blend=blend)
    torch.cuda.synchronize()


print("float32 device tensor  interleaved kernel: %7.3f ms   per-channel planes: %7.3f ms" % (best(lambda: dev(None)), best(lambda: dev("f32"))))
