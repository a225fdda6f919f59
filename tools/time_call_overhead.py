import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp
L = F.lib()
t = torch.rand((64, 64), device="cuda"); out = torch.empty_like(t)
fact = [1.0, 1e-3, 1e-6, 1e-9, 1e-12]
fa, n = F.fact_array(fact)
def best(fn, n=2000):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    return min(ts) * 1e6
print("raw ABI call, 64x64 device:            %.1f us" % best(lambda: L.dcp_unwarp_image_f32(t.data_ptr(), out.data_ptr(), 64, 64, 64, 1, 32.0, 32.0, fa, n, 1, 1, 1, 1, 0, None)))
print("pp.unwarp_image_backward(tensor):      %.1f us" % best(lambda: pp.unwarp_image_backward(t, 32.0, 32.0, fact)))
print("pp.unwarp_image_backward(tensor, out): %.1f us" % best(lambda: pp.unwarp_image_backward(t, 32.0, 32.0, fact, out=out)))
a = np.random.rand(64, 64).astype(np.float32)
print("pp.unwarp_image_backward(numpy 64x64): %.1f us" % best(lambda: pp.unwarp_image_backward(a, 32.0, 32.0, fact), 500))
