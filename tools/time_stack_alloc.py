"""Does the overlap of the streamed stack path depend on how the host arrays were allocated?"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
from discorpy_amd.post import postprocessing as pp
F.lib(); F.require_device()
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
c4 = configs.cfg4(16)
shape_in, shape_out = (16, 2560, 2560), (16, 2560, 2560)
n_in = int(np.prod(shape_in)) * 4
libc = ctypes.CDLL("libc.so.6")
libc.aligned_alloc.restype = ctypes.c_void_p
libc.aligned_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def best(fn, n=4):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


def arrays(kind):
    if kind == "numpy":
        return np.ones(shape_in, np.float32), np.zeros(shape_out, np.float32)
    out = []
    for _ in range(2):
        p = libc.aligned_alloc(1 << 21, n_in)
        if kind == "nohuge":
            libc.madvise(p, n_in, 15)       # MADV_NOHUGEPAGE
        elif kind == "huge":
            libc.madvise(p, n_in, 14)       # MADV_HUGEPAGE
        a = np.ctypeslib.as_array((ctypes.c_float * (n_in // 4)).from_address(p)).reshape(shape_in)
        a[:] = 1.0
        out.append(a)
    return out


for kind in ("numpy", "malloc", "nohuge", "huge"):
    vol, out = arrays(kind)
    t = best(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 0, 2559, out=out))
    print("%-8s 419 MB up + 419 MB down: %.2f ms" % (kind, t), flush=True)
