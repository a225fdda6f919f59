"""Per-wave phase timeline of one cfg2 launch of remap_lds_kernel (the lab build: `make -C discorpy_amd/csrc lab`, DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_lab.so):
DCP_LIB_PATH=discorpy_amd/lib/libdcp_var_trace.so python tools/trace_k1.py [blend order]"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discorpy_amd import _ffi as F, configs
L = F.lib(); F.require_device()
blend = int(sys.argv[1]) if len(sys.argv) > 1 else 1
order = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = configs.cfg2(); H, W = c["shape"]
rng = np.random.default_rng(c["seed"])
NR = 12
src = [F.DeviceBuffer(H * W * 4).upload(rng.random((H, W), dtype=np.float32)) for _ in range(NR)]
dst = [F.DeviceBuffer(H * W * 4) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])
def launch(k):
    F.check(L.dcp_unwarp_image_f32(src[k % NR].ptr, dst[k % NR].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, order, 1, blend, 1, -1, None))
for k in range(3000):
    launch(k)
F.check(L.dcp_stream_synchronize(-1, None))
e0, e1 = F.Event(), F.Event(); e0.record(); launch(5); e1.record(); e1.synchronize()
us = e0.elapsed_ms(e1) * 1e3
NW = 16384
buf = (C.c_uint64 * (NW * 12))()
L.dcp_experiment_read_trace.argtypes = [C.c_void_p, C.c_int]
assert L.dcp_experiment_read_trace(buf, NW) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 12).astype(np.int64)
hw = t[:, 7]
xcc = hw >> 32
dts = t[:, :7] - t[:, :1]                       # per-wave phase offsets in s_memtime ticks (shader clock)
rt0 = (t[:, 8] - t[:, 8].min()) * 10.0          # s_memrealtime: one 100 MHz counter for the whole chip -> ns
rt1 = (t[:, 9] - t[:, 8].min()) * 10.0
tick_ns = np.median((rt1 - rt0)[dts[:, 6] > 0] / dts[:, 6][dts[:, 6] > 0])
ts = rt0[:, None] + dts * tick_ns               # global timeline in ns
span = ts[:, 6].max()
print("launch %.2f us by events; first start -> last end %.2f us; one s_memtime tick = %.3f ns" % (us, span / 1e3, tick_ns))
tick_ns_phase = tick_ns
names = ["start->barrier", "P1a+box", "fill issue", "P1b", "fill wait", "P2"]
d = np.diff(dts, axis=1).astype(np.float64)
for i, nme in enumerate(names):
    print("%-16s mean %8.1f ticks (%6.3f us)  p10 %8.1f  p50 %8.1f  p90 %8.1f" % (nme, d[:, i].mean(), d[:, i].mean() * tick_ns / 1e3,
          np.percentile(d[:, i], 10), np.percentile(d[:, i], 50), np.percentile(d[:, i], 90)))
life = (dts[:, 6] - dts[:, 0]).astype(np.float64)
print("wave lifetime mean %.1f ticks = %.3f us" % (life.mean(), life.mean() * tick_ns / 1e3))
cu = (hw & 0xf00) >> 8; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print("hw_id fields: cu ids", np.unique(cu), "se", np.unique(se), "simd", np.unique(simd), "xcc", np.unique(xcc))
# timeline in 40 bins: waves resident, and in each phase
nb = 40
edges = np.linspace(0, span, nb + 1)
print("bin_us  resident  wait_barrier  P1a  fill  P1b  fillwait  P2   started  ended")
for b in range(nb):
    mid = 0.5 * (edges[b] + edges[b + 1])
    res = np.count_nonzero((ts[:, 0] <= mid) & (ts[:, 6] > mid))
    ph = [np.count_nonzero((ts[:, i] <= mid) & (ts[:, i + 1] > mid)) for i in range(6)]
    st = np.count_nonzero((ts[:, 0] >= edges[b]) & (ts[:, 0] < edges[b + 1]))
    en = np.count_nonzero((ts[:, 6] >= edges[b]) & (ts[:, 6] < edges[b + 1]))
    print("%6.2f  %7d  %s  %6d %6d" % (mid / 1e3, res, " ".join("%6d" % v for v in ph), st, en))
