"""Device-resident timing of whole-shard stack calls by element type: stack_wg_kernel<..., T> (option stack_wg = 1, the default)
against the generic one-thread-per-voxel typed_stack_kernel (stack_wg = 0).  cfg4 geometry, a 64-projection shard, every row.

    python tools/time_typed_stack.py [depth]
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from discorpy_amd import configs, _ffi as F
from discorpy_amd.post import postprocessing as pp

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c4 = configs.cfg4(D)
H = W = 2560


QUICK = bool(os.environ.get("TTS_QUICK"))           # one timed launch per case (counter passes: tools/pmc_typed_stack.sh)


def best(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    n = 1 if QUICK else n
    for _ in range(1 if QUICK else 3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n)
    return min(ts) * 1e6


for dt in (torch.float32, torch.uint16, torch.int32, torch.uint32, torch.float64):
    if dt == torch.uint32:
        vol = torch.from_numpy(np.random.default_rng(1).integers(0, 2**32 - 1, size=(D, H, W), dtype=np.uint32)).cuda()
    else:
        vol = (torch.rand((D, H, W), device="cuda") * 60000).to(dt)
    out = torch.empty((D, H, W), dtype=dt, device="cuda")
    res = {}
    for wg in (1, 0):
        F.set_option("x_stack_wg", wg)
        us = best(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 0, H - 1, out=out))
        res[wg] = (us, F.last_kernel(), out.clone() if wg else None)
        nb = out.numel() * out.element_size() * 2
        print("%-8s D=%d all rows stack_wg=%d: %9.1f us  %5.3f of 8 TB/s   %s" % (str(dt).replace("torch.", ""), D, wg, us, nb / us / 1e6 / 8.0, F.last_kernel()), flush=True)
    same = torch.equal(res[1][2].view(torch.uint8), out.view(torch.uint8))
    print("         staged == generic: %s" % same, flush=True)
    F.set_option("x_stack_wg", 1)
    del vol, out, res
