"""A/B of the launch options of stack_wg_kernel<..., 32-bit integer> on one box, one process: depth chunk, XCD tile order, store wait.
cfg4 geometry, one shard, every row.   python tools/ab_int32_stack.py [depth]"""
import sys, time
import torch
sys.path.insert(0, ".")
from discorpy_amd import configs, _ffi as F
from discorpy_amd.post import postprocessing as pp

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c4 = configs.cfg4(D)
H = W = 2560
DEFAULTS = {"x_d_chunk": 16, "x_xcd_remap": 2, "x_store_wait": 1}


def best(fn, n=6):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n)
    return min(ts) * 1e6


for dt in (torch.int32, torch.float32):
    vol = (torch.rand((D, H, W), device="cuda") * 60000).to(dt)
    out = torch.empty((D, H, W), dtype=dt, device="cuda")
    call = lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 0, H - 1, out=out)
    for rnd in range(2):
        for key, values in (("x_d_chunk", (16, 8, 32)), ("x_xcd_remap", (2, 1, 0)), ("x_store_wait", (1, 0))):
            for v in values:
                F.set_option(key, v)
                us = best(call)
                print("%-8s round %d %-10s = %2d: %8.1f us  %s" % (str(dt).replace("torch.", ""), rnd, key, v, us, F.last_kernel()), flush=True)
            F.set_option(key, DEFAULTS[key])
    del vol, out
