"""How independent per-frame launches are dispatched: cfg2, one dcp_unwarp_image_f32 call per frame,
ring of 24 frame pairs, the same process and box for every mode, modes alternated `--rounds` times.

    ordered        plain launches on one stream (the AQL barrier bit set: a frame starts when the previous one has drained)
    any_order      the same calls with DCP_MEM_DEVICE_UNORDERED (hipExtAnyOrderLaunch: barrier bit cleared)
    two_streams    frames alternated over two library streams, ordered inside each (streamsN: over N streams, N = 3..8)
    two_any        both
    batch          dcp_unwarp_images_f32, 24 frames with 24 calibrations per call (remap_wg_batch_kernel: the reference point)

Times: host wall clock between two full synchronisations around `--reps` launches (and HIP events on the stream where one
stream carries everything).  Every mode's frames are compared with the ordered result once.
    python tools/time_dispatch.py [--reps 1920] [--rounds 3] [--blend 1] [--order 1]"""
import argparse
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from discorpy_amd import _ffi as F, configs

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=1920)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--blend", type=int, default=1)
ap.add_argument("--order", type=int, default=1)
ap.add_argument("--modes", default="ordered,any_order,two_streams,two_any,batch")
a = ap.parse_args()

L = F.lib(); F.require_device()
c = configs.cfg2(); H, W = c["shape"]
rng = np.random.default_rng(c["seed"])
NR = 24
src = [F.DeviceBuffer(H * W * 4).upload(rng.random((H, W), dtype=np.float32)) for _ in range(NR)]
dst = [F.DeviceBuffer(H * W * 4) for _ in range(NR)]
fa, n = F.fact_array(c["list_fact"])
streams = [F.Stream() for _ in range(8)]
s0, s1 = streams[0], streams[1]
UNORDERED = getattr(F, "MEM_DEVICE_UNORDERED", 0x101)


def one(k, mem_kind, stream):
    F.check(L.dcp_unwarp_image_f32(src[k].ptr, dst[k].ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, n, a.order, 1, a.blend, mem_kind, -1, stream))


srcs = (C.c_void_p * NR)(*[b.ptr for b in src]); dsts = (C.c_void_p * NR)(*[b.ptr for b in dst])
xcs = (C.c_double * NR)(*([c["xcenter"]] * NR)); ycs = (C.c_double * NR)(*([c["ycenter"]] * NR))
# 24 DIFFERENT calibrations would route to the batch kernel; equal ones go to the stack kernel -- perturb the centres in the last bit
xcs = (C.c_double * NR)(*[c["xcenter"] + 1e-9 * i for i in range(NR)])
facts = (C.c_double * (NR * n))(*(list(c["list_fact"]) * NR))


def sync():
    for s_ in streams:
        s_.synchronize()


def run(mode, reps):
    sync()
    e0, e1 = F.Event(), F.Event()
    single = mode in ("ordered", "any_order", "batch")
    t0 = time.perf_counter()
    if single:
        e0.record(s0.ptr)
    if mode == "batch":
        for r in range(reps // NR):
            F.check(L.dcp_unwarp_images_f32(srcs, dsts, NR, H, W, W, 1, xcs, ycs, facts, n, a.order, 1, a.blend, 1, -1, s0.ptr))
    else:
        mk = UNORDERED if mode in ("any_order", "two_any") else 1
        ns = 2 if mode in ("two_streams", "two_any") else int(mode[7:]) if mode.startswith("streams") else 1
        for r in range(reps):
            one(r % NR, mk, streams[r % ns].ptr)
    t_enq = time.perf_counter()
    if single:
        e1.record(s0.ptr)
    sync()
    t1 = time.perf_counter()
    ev = e0.elapsed_ms(e1) / reps * 1e3 if single else float("nan")
    return (t1 - t0) / reps * 1e6, ev, (t_enq - t0) / reps * 1e6


modes = a.modes.split(",")
# correctness: every mode's ring equals the ordered ring
ref = None
for m in ["ordered"] + [m for m in modes if m != "ordered"]:
    for b in dst:
        b.upload(np.zeros((H, W), np.float32))
    run(m, NR)
    got = [b.download((H, W), np.float32) for b in dst]
    if m == "batch":
        continue           # other centres by construction
    if ref is None:
        ref = got
    else:
        bad = sum(int((g.view(np.uint32) != r.view(np.uint32)).sum()) for g, r in zip(got, ref))
        print("check %-12s differing values vs ordered: %d  (%s)" % (m, bad, F.last_kernel()), flush=True)

# settle the clocks
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    run("ordered", NR * 4)
for rnd in range(a.rounds):
    for m in modes:
        wall, ev, enq = run(m, a.reps)
        print("round %d  %-12s wall %7.2f us/frame  events %7.2f  host enqueue %5.2f  frac(wall) %.3f  %s" %
              (rnd, m, wall, ev, enq, 8.0 * H * W / wall / 1e6 / 8.0, F.last_kernel()), flush=True)
