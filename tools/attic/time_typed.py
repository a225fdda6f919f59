"""Device-resident timings of the element-type kernels (typed_kernels.hip) next to the float32 hot path."""
import sys, time
import torch
sys.path.insert(0, ".")
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp
c = configs.cfg2(); c4 = configs.cfg4(64)


def best(fn, n=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n)
    return min(ts) * 1e6


for dt in (torch.float32, torch.float64, torch.uint8, torch.uint16, torch.int32):
    img = (torch.rand((4096, 4096), device="cuda") * 200).to(dt)
    out = torch.empty_like(img)
    for order in (1, 0):
        us = best(lambda: pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], order=order, out=out))
        nb = img.element_size() * 2 * img.numel()
        print("%-8s 4096^2 order %d: %8.1f us  %6.2f TB/s algorithmic" % (str(dt).replace("torch.", ""), order, us, nb / us / 1e6), flush=True)
for dt in (torch.float32, torch.uint16):
    vol = (torch.rand((64, 2560, 2560), device="cuda") * 60000).to(dt)
    out = torch.empty((64, 64, 2560), dtype=dt, device="cuda")
    us = best(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1000, 1063, out=out))
    print("%-8s stack D=64, 64 rows: %8.1f us  %6.2f TB/s algorithmic" % (str(dt).replace("torch.", ""), us, out.numel() * out.element_size() * 2 / us / 1e6), flush=True)
