"""Where the NumPy -> NumPy time of one cfg2 frame goes: page faults of a fresh output, H2D, kernel, D2H."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
from discorpy_amd.post import postprocessing as pp
L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]
img = np.random.default_rng(1).random((H, W), dtype=np.float32)
fa, n = F.fact_array(c["list_fact"])


def best(fn, n=9):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


def call(out):
    F.check(L.dcp_unwarp_image_f32(img.ctypes.data, out.ctypes.data, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, 1, 1, 1, 0, -1, None))


def fresh_touch():
    a = np.empty((H, W), np.float32); a.reshape(-1)[::1024] = 0; return a


pre = np.zeros((H, W), np.float32)
print("np.empty + first touch of every page (64 MiB): %.2f ms" % best(fresh_touch))
print("ABI call, output already faulted in:            %.2f ms" % best(lambda: call(pre)))
print("ABI call, fresh np.empty output each time:      %.2f ms" % best(lambda: call(np.empty((H, W), np.float32))))
print("pp.unwarp_image_backward (fresh output):        %.2f ms" % best(lambda: pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"])))
d_src, d_dst = F.DeviceBuffer(img.nbytes), F.DeviceBuffer(img.nbytes)
print("H2D alone (dcp_memcpy):                         %.2f ms" % best(lambda: d_src.upload(img)))
print("D2H alone into faulted-in array:                %.2f ms" % best(lambda: F.check(L.dcp_memcpy(pre.ctypes.data, d_dst.ptr, pre.nbytes, F.COPY_D2H, -1, None))))
