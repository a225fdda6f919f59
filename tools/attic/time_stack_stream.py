"""Host (NumPy) stack through the streamed DCP_MEM_HOST path for several chunk sizes."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs
from discorpy_amd.post import postprocessing as pp
F.lib(); F.require_device()
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vol = np.random.default_rng(4).random((depth, 2560, 2560), dtype=np.float32)
c4 = configs.cfg4(depth)
out = np.zeros((depth, 64, 2560), np.float32)


def best(fn, n=4):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


for kb in (2048, 8192, 24576, 65536, 262144, 4194304):
    F.set_option("stack_chunk_kb", kb)
    t = best(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1000, 1063, out=out))
    print("chunk %7d KiB: 64 rows x depth %d numpy->numpy %.2f ms  (up %.0f MB, down %.0f MB)" % (kb, depth, t, depth * 70 * 2560 * 4 / 1e6, out.nbytes / 1e6), flush=True)
F.set_option("stack_chunk_kb", 24576)
full = np.zeros((16, 2560, 2560), np.float32)
t = best(lambda: pp.unwarp_chunk_slices_backward(vol[:16], c4["xcenter"], c4["ycenter"], c4["list_fact"], 0, 2559, out=full), 3)
print("all 2560 rows of 16 projections (419 MB up, 419 MB down): %.2f ms = %.1f GB/s each way" % (t, full.nbytes / t / 1e6))
