#!/usr/bin/env python
"""bench.py -- Mpixels/s of the backward unwarp on MI355X (BASELINE.json metric).

Headline at every N: BASELINE config 2 -- 4096x4096 float32 frames, 5-term backward polynomial
(coef_dot_05 rescaled), bilinear -- device-resident, ONE launch of dcp_unwarp_image_f32 per frame.  One "step"
is `cycles` passes of the hot path over a ring of `--batch` DISTINCT frames (default 24: 3.2 GB of input+output
per GPU, so the 256 MiB Infinity Cache cannot hold the working set); `cycles` is chosen from one probe pass so that
the K timed steps last at least --min-timed-ms (300 ms) whatever K is, and the line says how many frames a step was
(config.frames_per_step_per_gpu).  `value` is computed from the HIP events on the launch stream around the K steps,
`value_wall` / `ms_per_step` from the host clock between the barrier + synchronize pairs, and the host's share is
spelled out (host_enqueue_us_per_launch, sync_ms).  With N > 1 every rank unwarps its own ring (independent frames,
no data-path collective): weak scaling.

After the timed region the same run also measures, outside it and reported beside the headline:
  other_configs   (N = 1) config 2 with scipy's exact blend and at order 0, config 3 fused / two-pass /
                  perspective only, config 5, config 4 on one GPU (whole stack when it fits) and one sinogram --
                  each with its per-launch time from HIP events, the kernel that ran and verified_vs_oracle;
  stack_scaling   (every N) config 4 sharded by depth over the ranks: compute only, and compute + the RCCL
                  all-gather that reassembles the (depth, rows, width) block on every rank -- the two curves
                  north_star asks for fall out of the driver's N = 1, 2, 4, 8 runs without a flag.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the
dominant kernel and `cpu_baseline` (oracle/unwarp_oracle.c timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from discorpy_amd import _ffi as F  # noqa: E402
from discorpy_amd import configs  # noqa: E402

BLEND_NAMES = {"scipy": F.BLEND_SCIPY, "f64lerp": F.BLEND_F64LERP, "f32lerp": F.BLEND_F32LERP}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pipeline", type=int, default=1,
                    help="stack workload: all-gather pipelined against the kernel in this many depth sub-blocks")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="back-to-back launches for this long before the warm-up steps: the core clock needs ~100 ms under "
                         "load to reach its sustained level (tools/time_ramp.py); 0 disables")
    ap.add_argument("--batch", type=int, default=24, help="distinct frames in the ring (3.2 GB at 24: the Infinity Cache cannot hold it)")
    ap.add_argument("--cycles", type=int, default=0,
                    help="passes over the ring per step; 0 = chosen from one probe pass so that the K timed steps last at least "
                         "--min-timed-ms whatever K is (a step is then cycles x batch frames, reported as frames_per_step_per_gpu)")
    ap.add_argument("--min-timed-ms", type=float, default=300.0,
                    help="lower bound of the timed region used to choose --cycles 0: a 14 ms region (20 steps x 24 frames) lost 25 %% "
                         "to 4.7 ms of one-off host latency on the driver's box in round 4")
    ap.add_argument("--dispatch", default="two_streams", choices=["two_streams", "ordered", "any_order"],
                    help="how the independent per-frame launches of the headline leave the host: two_streams (default) alternates the "
                         "ring's frames over two streams of the library (dcp_stream_create), so that two frames are in flight and the "
                         "drain of one runs under the ramp of the other; ordered = one stream, every launch behind the previous one's "
                         "last workgroup (rounds 1-5); any_order = one stream, DCP_MEM_DEVICE_UNORDERED (barrier bit cleared).  "
                         "profiles/r06a_dispatch_modes.txt, tools/time_dispatch.py")
    ap.add_argument("--blend", default="f64lerp", choices=sorted(BLEND_NAMES))
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip other_configs and stack_scaling")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the CPUs this process may use (affinity, cgroup quota)")
    ap.add_argument("--option", action="append", default=[], help="kernel option key=value (dcp_set_option)")
    ap.add_argument("--workload", default="frame", choices=["frame", "stack"],
                    help="frame: BASELINE config 2 (default, the headline metric); stack: config 4 as the metric, the "
                         "(depth, 2560, 2560) stack sharded over the ranks by depth + all-gather")
    ap.add_argument("--depth", type=int, default=2048, help="stack: total number of projections")
    ap.add_argument("--rows", type=int, default=2560, help="stack: output rows per step (2560 = whole stack)")
    ap.add_argument("--no-gather", action="store_true", help="stack workload: skip the all-gather")
    ap.add_argument("--e2e-child", action="store_true", help=argparse.SUPPRESS)   # internal: the NumPy -> NumPy timings, in a process of their own
    # internal: the torch-free exchange variants of config 4 (C ABI only), each in processes of their own (native_exchange_variants)
    ap.add_argument("--native-child", default=None, choices=["rccl", "peer"], help=argparse.SUPPRESS)
    ap.add_argument("--child-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--child-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--idfile", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-spawn", action="store_true",
                    help="--gpus N > 1 without WORLD_SIZE in the environment normally re-launches itself as N ranks; this keeps one process")
    ap.add_argument("--shard", default="depth", choices=["depth", "rows"],
                    help="stack workload: depth = projections split over the ranks (+ all-gather); rows = every rank owns "
                         "output rows of every projection (the whole stack on every rank, no collective)")
    return ap.parse_args(argv)


def kernel_sources_digest():
    """SHA-256 over the sources the library is built from (discorpy_amd/csrc): what ties a committed PMC figure
    (profiles/pmc_latest.json, written by tools/profile.sh on the GPU box) to the code that is being timed."""
    import hashlib
    d = os.path.join(ROOT, "discorpy_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp", ".map")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def usable_cpus():
    """CPUs this process may really use: the smaller of the affinity mask and the cgroup quota (the GPU boxes
    show 256 logical CPUs but grant 16; running 64 threads there is slower than running 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            quota = None
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def oracle_module(threads=0):
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(threads if threads > 0 else min(orc.max_threads(), usable_cpus()))
    return orc


def cpu_baseline(cfg, img, blend, threads):
    """Time the oracle (a C port of the reference arithmetic) on the host: whole 4096^2 frames."""
    orc = oracle_module(threads)
    ncores = orc.max_threads()
    t = threads if threads > 0 else min(ncores, usable_cpus())
    kw = dict(order=cfg["order"], poly=orc.POLY_NUMPY, blend=orc.BLEND_SCIPY)
    orc.unwarp_image_backward(img, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], **kw)  # warm
    frames, t0 = 0, time.perf_counter()
    while True:
        orc.unwarp_image_backward(img, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], **kw)
        frames += 1
        dt = time.perf_counter() - t0
        if dt * t >= 20.0 or frames >= 64:   # ~20 CPU-seconds of work
            break
    mpix = frames * img.size / dt / 1e6
    return {"value": round(mpix, 2), "unit": "Mpixels/s", "cores": t, "kind": "port",
            "sample": "%d full %dx%d frames of the bench workload, reference arithmetic order "
                      "(numpy-order polynomial, scipy blend), %d OpenMP threads (%d logical CPUs visible, %d usable "
                      "under the affinity mask / cgroup quota), %.2f s wall.  The reference itself (Python: numpy + scipy, it "
                      "cannot use more than one core and may not travel to this box) measured on one core of the build container: "
                      "1.97 s per frame = 8.5 Mpixels/s (profiles/rounds_1-4/r01c_reference_cpu_build_container.txt, tools/time_reference.py)"
                      % (frames, img.shape[0], img.shape[1], t, ncores, usable_cpus(), dt)}


# ----------------------------------------------------------------------------------------- timing helpers

# How INDEPENDENT per-frame launches leave the host (--dispatch): the headline and every other one-launch-per-frame entry use the
# same mode.  two_streams: launch i goes to stream i & 1 of two library streams (dstream), timed regions fork from / join into the
# first of them (timed_launches(..., dispatch=True)).  Entries whose launches depend on each other (the two-pass chain,
# stacks, batches) stay on the default stream.
_DISPATCH = {"mode": "ordered", "streams": [], "join": None}


def set_dispatch(mode, dev):
    _DISPATCH["mode"] = mode
    _DISPATCH["streams"] = [F.Stream(dev), F.Stream(dev)] if mode == "two_streams" else []
    _DISPATCH["join"] = F.Event(dev) if mode == "two_streams" else None


def dstream(i):
    st = _DISPATCH["streams"]
    return st[i & 1].ptr if st else None


def dmem():
    return F.MEM_DEVICE_UNORDERED if _DISPATCH["mode"] == "any_order" else F.MEM_DEVICE


def dsync(dev):
    for st in _DISPATCH["streams"]:
        st.synchronize()
    F.check(F.lib().dcp_stream_synchronize(dev, None))


def timed_launches(fn, reps, dev, settle_ms=120.0, dispatch=False):
    """Average device time of fn(i) per call: HIP events on the launch stream around `reps` back-to-back calls, after
    `settle_ms` of the same calls (clock ramp).  dispatch=True: fn(i) launches on dstream(i) -- under two_streams the start event is
    recorded on the first stream and the second waits for it, the stop event on the first behind an event of the second."""
    L = F.lib()
    st = _DISPATCH["streams"] if dispatch else []
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < settle_ms or i < 3:
        fn(i)
        i += 1
        if i % 64 == 0:
            dsync(dev) if st else F.check(L.dcp_stream_synchronize(dev, None))
    dsync(dev) if st else F.check(L.dcp_stream_synchronize(dev, None))
    e0, e1 = F.Event(dev), F.Event(dev)
    if st:
        e0.record(st[0].ptr)
        st[1].wait_event(e0)
    else:
        e0.record()
    for r in range(reps):
        fn(r)
    if st:
        _DISPATCH["join"].record(st[1].ptr)
        st[0].wait_event(_DISPATCH["join"])
        e1.record(st[0].ptr)
    else:
        e1.record()
    e1.synchronize()
    return e0.elapsed_ms(e1) * 1e3 / reps


def launch_distribution(fn, n, dev):
    """Device time of each of n consecutive fn(i) launches (one HIP event between every two): median, mean and tail.  The
    intervals include the gap to the next launch, as the headline's average does."""
    L = F.lib()
    ev = [F.Event(dev) for _ in range(n + 1)]
    t_settle, i = time.perf_counter(), 0
    while (time.perf_counter() - t_settle) * 1e3 < 250.0:       # the device has idled through the checks before this: let the clock settle again
        fn(i)
        i += 1
        if i % 64 == 0:
            F.check(L.dcp_stream_synchronize(dev, None))
    F.check(L.dcp_stream_synchronize(dev, None))
    for i in range(n):
        ev[i].record()
        fn(i)
    ev[n].record()
    ev[n].synchronize()
    us = np.array([ev[i].elapsed_ms(ev[i + 1]) * 1e3 for i in range(n)])
    # what the event between two launches costs by itself (a marker packet with a cache release: the next kernel starts later
    # than it would behind another kernel): the same kernel in pairs, one event per pair -- (pair - 2 x back-to-back) is not
    # observable directly, so report the back-to-back average beside the per-interval figures and let the reader subtract
    e0, e1 = F.Event(dev), F.Event(dev)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    b2b = e0.elapsed_ms(e1) * 1e3 / n
    return {"launches": n, "median_us": round(float(np.median(us)), 3), "mean_us": round(float(us.mean()), 3),
            "p10_us": round(float(np.percentile(us, 10)), 3), "p90_us": round(float(np.percentile(us, 90)), 3),
            "max_us": round(float(us.max()), 3), "back_to_back_mean_us": round(b2b, 3),
            "event_overhead_us": round(float(np.median(us)) - b2b, 3),
            "note": "each interval = one launch + one HIP event record between launches; back_to_back_mean_us is the same n launches with "
                    "no events in between, event_overhead_us = median_us - back_to_back_mean_us"}


def hip_runtime_path():
    try:
        for ln in open("/proc/self/maps"):
            if "libamdhip64" in ln:
                return ln.split()[-1]
    except OSError:
        pass
    return None


def e2e_child():
    """NumPy in -> NumPy out through the drop-in functions (what a caller of discorpy.post.postprocessing with host arrays
    sees: PCIe-inclusive, SURVEY.md section 8(d) "secondary").  Runs in a process of its own so that the parent can time both
    HIP runtimes (the one bundled with PyTorch-ROCm, which the library shares with torch by default, and /opt/rocm's)."""
    from discorpy_amd.post import postprocessing as pp
    F.require_device()
    inherited = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    try:            # the parent's OpenMP runtime may have pinned its main thread: this process moves data on two host threads
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except (AttributeError, OSError):
        pass
    res = {"hip_runtime": hip_runtime_path(), "cpus_inherited": inherited,
           "cpus_now": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
           "blend": pp._default_blend(True) + " (blend=None on NumPy arrays: scipy's exact operation order, the reference's result bit for bit)"}
    c2 = configs.cfg2()
    img = np.random.default_rng(c2["seed"]).random(c2["shape"], dtype=np.float32)

    groups = {}

    def med(fn, reps, name=None):
        """Median of `reps` calls -- the LOWEST of three such groups half a second apart: the boxes of the pool are shared, and a
        neighbour's burst on the host's memory or PCIe complex lasts seconds (seen: a whole group at 4.4-4.8 ms between groups at
        1.8 ms); every group's median is kept in `group_medians_ms`."""
        fn()
        fn()
        meds = []
        for g in range(3):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            meds.append(float(np.median(ts)) * 1e3)
            if g < 2:
                time.sleep(0.5)
        if name:
            groups[name] = [round(v, 3) for v in meds]
        return min(meds)
    ms = med(lambda: pp.unwarp_image_backward(img, c2["xcenter"], c2["ycenter"], c2["list_fact"]), 9, "cfg2")
    res["cfg2_unwarp_image_backward_4096"] = {"ms": round(ms, 3), "Mpixels_per_s": round(img.size / ms / 1e3, 1), "group_medians_ms": groups["cfg2"]}
    c4 = configs.cfg4(32)
    D, H, W = c4["shape"]
    vol = np.random.default_rng(c4["seed"]).random((D, H, W), dtype=np.float32)
    ms = med(lambda: pp.unwarp_slice_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1277), 12)
    res["cfg4_unwarp_slice_backward_depth32"] = {"ms": round(ms, 3), "Mpixels_per_s": round(D * W / ms / 1e3, 2)}
    ms = med(lambda: pp.unwarp_chunk_slices_backward(vol, c4["xcenter"], c4["ycenter"], c4["list_fact"], 1248, 1311), 6)
    res["cfg4_unwarp_chunk_slices_64rows_depth32"] = {"ms": round(ms, 3), "Mpixels_per_s": round(D * 64 * W / ms / 1e3, 1)}
    try:
        res["pcie"] = pcie_floor(img.nbytes)
        fl = res["pcie"].get("both_directions_concurrent_ms")
        if fl:
            res["cfg2_unwarp_image_backward_4096"]["pcie_floor_ms"] = fl
            res["cfg2_unwarp_image_backward_4096"]["of_pcie_floor"] = round(res["cfg2_unwarp_image_backward_4096"]["ms"] / fl, 3)
    except Exception as e:      # noqa: BLE001 -- context only
        res["pcie"] = {"error": repr(e)}
    print(json.dumps(res), flush=True)


def pcie_floor(nbytes, reps=7):
    """What the PCIe link of THIS box and THIS HIP runtime gives a frame's worth of data: `nbytes` up, `nbytes` down, from / into
    registered (pinned, page-aligned) host memory through the C ABI's copy helper -- each direction alone, one after the other, and
    both at once from two host threads on two streams.  `both_directions_concurrent_ms` is the floor of a NumPy -> NumPy call that
    moves its frame with runtime copies; where a runtime does not overlap the two (the one bundled with PyTorch-ROCm), it equals
    the serial figure and the drop-in's direct-write path is what beats it."""
    import ctypes as C
    import mmap
    import threading
    L = F.lib()
    dev = -1
    bufs = [mmap.mmap(-1, nbytes) for _ in range(2)]          # page-aligned, nothing else in their pages
    ptrs = [C.addressof(C.c_char.from_buffer(b)) for b in bufs]
    for b in bufs:
        b.write(b"\1" * nbytes)
    for p_ in ptrs:
        F.check(L.dcp_host_register(p_, nbytes, dev))
    d_up, d_down = F.DeviceBuffer(nbytes, dev), F.DeviceBuffer(nbytes, dev)
    s_up, s_down = C.c_void_p(), C.c_void_p()
    F.check(L.dcp_stream_create(C.byref(s_up), dev))
    F.check(L.dcp_stream_create(C.byref(s_down), dev))

    def up():
        F.check(L.dcp_memcpy(d_up.ptr, ptrs[0], nbytes, F.COPY_H2D, dev, s_up))

    def down():
        F.check(L.dcp_memcpy(ptrs[1], d_down.ptr, nbytes, F.COPY_D2H, dev, s_down))

    def both():
        th = threading.Thread(target=down)
        th.start()
        up()
        th.join()

    def serial():
        up()
        down()

    def best(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(min(ts) * 1e3, 3)
    try:
        out = {"bytes_each_way": int(nbytes), "h2d_ms": best(up), "d2h_ms": best(down), "both_directions_serial_ms": best(serial),
               "both_directions_concurrent_ms": best(both)}
        out["runtime_overlaps_directions"] = bool(out["both_directions_concurrent_ms"] < 0.85 * out["both_directions_serial_ms"])
        out["h2d_GBps"], out["d2h_GBps"] = round(nbytes / out["h2d_ms"] / 1e6, 1), round(nbytes / out["d2h_ms"] / 1e6, 1)
    finally:
        for p_ in ptrs:
            L.dcp_host_unregister(p_)
        L.dcp_stream_destroy(s_up)
        L.dcp_stream_destroy(s_down)
        d_up.free()
        d_down.free()
    return out


def end_to_end_numpy():
    """The e2e child under both HIP runtimes (when torch is installed its bundled runtime is the library's default)."""
    import subprocess
    out = {"what": "NumPy in -> NumPy out through discorpy_amd.post.postprocessing (host arrays: H2D + kernel + D2H, median of "
                   "repeated calls); the reference on one core: cfg2 1965 ms, slice of a depth-64 stack 6.7 ms "
                   "(profiles/rounds_1-4/r01c_reference_cpu_build_container.txt)"}
    for label, env in (("default_runtime", {}), ("system_rocm_runtime", {"DISCORPY_AMD_SYSTEM_HIP": "1"})):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--e2e-child"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, **env), cwd=ROOT)
            lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
            out[label] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as e:      # noqa: BLE001 -- context only
            out[label] = {"error": repr(e)}
    a_, b_ = out.get("default_runtime", {}), out.get("system_rocm_runtime", {})
    if a_.get("hip_runtime") and a_.get("hip_runtime") == b_.get("hip_runtime"):
        out["note"] = "both runs loaded the same libamdhip64.so (no second runtime on this box)"
    return out


def entry(us, pixels, bytes_per_pixel, kernel, verified, **more):
    gbps = bytes_per_pixel * pixels / (us * 1e-6) / 1e9
    d = {"launch_us": round(us, 3), "Mpixels_per_s": round(pixels / us, 1), "achieved_GBps": round(gbps, 1),
         "frac": round(gbps / configs.HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_pixel": bytes_per_pixel, "kernel": kernel,
         "verified_vs_oracle": bool(verified)}
    d.update(more)
    return d


def pf_entry(*args, **more):
    """entry() of a one-launch-per-independent-frame workload: timed under the bench's dispatch mode."""
    return entry(*args, dispatch=_DISPATCH["mode"], **more)


class RingFrame:
    """One frame of the ring: a view (pointer + length) into the ring's single allocation."""

    def __init__(self, ring, offset):
        self.ptr, self.nbytes, self.device = ring.ptr + offset, ring.nbytes - offset, ring.device

    def upload(self, array):
        a = np.ascontiguousarray(array)
        F.check(F.lib().dcp_memcpy(self.ptr, a.ctypes.data, a.nbytes, F.COPY_H2D, self.device, None))
        return self


def download(buf_ptr, shape, dev, offset=0):
    out = np.empty(shape, np.float32)
    F.check(F.lib().dcp_memcpy(out.ctypes.data, buf_ptr + offset, out.nbytes, F.COPY_D2H, dev, None))
    return out


def clocks_under_load(step, sync):
    """Core clock and package power while the headline kernel runs back to back (rocm-smi sampled from this thread while a second
    thread keeps launching): boxes of the pool differ by ~10 % in what they sustain, and the number says which kind this run got."""
    import re
    import subprocess
    import threading
    stop = threading.Event()

    def spin():
        while not stop.is_set():
            step()
            sync()
    th = threading.Thread(target=spin, daemon=True)
    th.start()
    samples = []
    try:
        time.sleep(0.35)
        for _ in range(2):
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\): ([0-9.]+)", txt)
            if m and w:
                samples.append((int(m.group(1)), float(w.group(1))))
    except Exception:      # noqa: BLE001 -- context only
        pass
    finally:
        stop.set()
        th.join()
    if not samples:
        return None
    return {"sclk_MHz_under_this_kernel": [s_[0] for s_ in samples], "package_power_W": [s_[1] for s_ in samples]}


def distinct_calibrations(cfg, n):
    """n calibrations around BASELINE config 2's (centre moved by up to ~100 px, coefficients rescaled by up to 4 %): all hold the
    level-2 tile certificate on 4096^2 frames."""
    cals = []
    for i in range(n):
        t = (i - 0.5 * (n - 1)) / max(n - 1, 1)
        cals.append((cfg["xcenter"] + 200.0 * t, cfg["ycenter"] - 120.0 * t,
                     [v * (1.0 + 0.04 * t * k) for k, v in enumerate(cfg["list_fact"])]))
    return cals


def batched_distinct_calibrations(a, dev, srcs, dsts, img0, cfg, blend):
    import ctypes as C
    L = F.lib()
    H, W = cfg["shape"]
    n = len(srcs)
    cals = distinct_calibrations(cfg, n)
    nf = len(cfg["list_fact"])
    table = np.ascontiguousarray([c[2] for c in cals], dtype=np.float64)
    sp = (C.c_void_p * n)(*[b.ptr for b in srcs])
    dp = (C.c_void_p * n)(*[b.ptr for b in dsts])
    xa, ya = (C.c_double * n)(*[c[0] for c in cals]), (C.c_double * n)(*[c[1] for c in cals])
    tp = table.ctypes.data_as(C.POINTER(C.c_double))

    def run(_i):
        F.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xa, ya, tp, nf, a.order, 1, blend, F.MEM_DEVICE, dev, None))
    per_frame_us = timed_launches(run, max(6, a.steps // 2), dev) / n
    kernel = F.last_kernel()
    run(0)
    orc = oracle_module(a.cpu_threads)
    ob = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32lerp": orc.BLEND_F32LERP}[a.blend]
    ok = True
    host = {0: img0}
    for i in sorted({0, n // 2, n - 1}):       # three frames with three different calibrations, every pixel
        if i not in host:
            host[i] = download(srcs[i].ptr, (H, W), dev)
        want = orc.unwarp_image_backward(host[i], cals[i][0], cals[i][1], cals[i][2], order=a.order, poly=orc.POLY_KERNEL, blend=ob)
        ok = ok and bool(np.array_equal(download(dsts[i].ptr, (H, W), dev), want))
    return {"what": "the %d frames of a step in ONE dcp_unwarp_images_f32 call, every frame with its own centre and coefficients "
                    "(blockIdx.z = frame)" % n,
            "us_per_frame": round(per_frame_us, 3), "Mpixels_per_s": round(H * W / per_frame_us, 1),
            "frac_of_hbm_peak": round(configs.BYTES_PER_PIXEL * H * W / (per_frame_us * 1e-6) / 1e9 / configs.HBM_PEAK_GBPS, 4),
            "kernel": kernel, "verified_vs_oracle": ok, "frames_checked": sorted({0, n // 2, n - 1})}


# ----------------------------------------------------------------------------------------- other configurations (N = 1)

def other_configs(a, dev, srcs, dsts, img0):
    """Per-launch times of the other BASELINE configurations and blend variants, each checked against the oracle on one
    frame (outside any timed region).  `srcs` / `dsts` / `img0`: the headline's ring of 4096^2 frames and the host copy of
    frame 0."""
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    out = {}
    c2, c3 = configs.cfg2(), configs.cfg3()
    H, W = c2["shape"]
    fa, nf = F.fact_array(c2["list_fact"])
    ca, _ = F.fact_array(c3["list_coef"])
    nring = len(srcs)
    reps = max(48, min(480, a.steps * 4))

    def radial(i, order, blend):
        F.check(L.dcp_unwarp_image_f32(srcs[i % nring].ptr, dsts[i % nring].ptr, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf,
                                       order, 1, blend, dmem(), dev, dstream(i)))

    def persp(i, src, dst):
        F.check(L.dcp_perspective_image_f32(src[i % nring].ptr, dst[i % nring].ptr, H, W, W, 1, ca, 1, F.BLEND_F64LERP, dmem(),
                                            dev, dstream(i)))

    def fused(i):
        F.check(L.dcp_unwarp_fused_f32(srcs[i % nring].ptr, dsts[i % nring].ptr, H, W, W, 1, c3["xcenter"], c3["ycenter"], fa, nf,
                                       ca, 1, F.BLEND_F64LERP, dmem(), dev, dstream(i)))

    a2 = (img0, c2["xcenter"], c2["ycenter"], c2["list_fact"])
    # config 2, scipy's exact operation order (bit-equal to the reference; the headline's f64lerp is <= 1 float32 ulp from it)
    us = timed_launches(lambda i: radial(i, 1, F.BLEND_SCIPY), reps, dev, dispatch=True)
    k = F.last_kernel()
    radial(0, 1, F.BLEND_SCIPY)
    dsync(dev)
    ok = np.array_equal(download(dsts[0].ptr, (H, W), dev), orc.unwarp_image_backward(*a2, poly=orc.POLY_KERNEL, blend=orc.BLEND_SCIPY))
    out["cfg2_scipy_exact_blend"] = pf_entry(us, H * W, 8, k, ok)
    us = timed_launches(lambda i: radial(i, 0, F.BLEND_SCIPY), reps, dev, dispatch=True)
    k = F.last_kernel()
    radial(0, 0, F.BLEND_SCIPY)
    dsync(dev)
    ok = np.array_equal(download(dsts[0].ptr, (H, W), dev), orc.unwarp_image_backward(*a2, order=0, poly=orc.POLY_KERNEL))
    out["cfg2_order0_nearest"] = pf_entry(us, H * W, 8, k, ok)
    # config 3: homography fused with the radial map in one resampling (north star), and what the reference does (two passes)
    us = timed_launches(fused, reps, dev, dispatch=True)
    k = F.last_kernel()
    fused(0)
    dsync(dev)
    ok = np.array_equal(download(dsts[0].ptr, (H, W), dev),
                        orc.unwarp_fused(img0, c3["xcenter"], c3["ycenter"], c3["list_fact"], c3["list_coef"], poly=orc.POLY_KERNEL,
                                         blend=orc.BLEND_F64LERP))
    out["cfg3_fused"] = pf_entry(us, H * W, 8, k, ok)
    us = timed_launches(lambda i: persp(i, srcs, dsts), reps, dev, dispatch=True)
    k = F.last_kernel()
    persp(0, srcs, dsts)
    dsync(dev)
    ok = np.array_equal(download(dsts[0].ptr, (H, W), dev), orc.correct_perspective_image(img0, c3["list_coef"], blend=orc.BLEND_F64LERP))
    out["cfg3_perspective_only"] = pf_entry(us, H * W, 8, k, ok)

    # order 3 (scipy's prefiltered cubic B-spline, the reference's `order` argument; mode "reflect"): the prefilter of both axes
    # in one launch (the column-filtered plane stays in LDS) + the LDS-staged 16-tap gather.  Algorithmic bytes: 4 read + 4
    # written per pixel, as for order 1 -- the float64 coefficient plane in between is the algorithm's own traffic.
    def cubic(i):
        F.check(L.dcp_unwarp_image_spline_f32(srcs[i % nring].ptr, dsts[i % nring].ptr, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf,
                                              3, 0, F.MEM_DEVICE, dev, dstream(i)))
    # (two streams: every stream keeps its own coefficient planes -- the library holds two workspaces per device --, so the prefilter of
    # one frame, memory-bound, runs under the gather of the other, LDS- and VALU-bound)
    us = timed_launches(cubic, max(24, reps // 4), dev, dispatch=True)
    k = F.last_kernel()
    cubic(0)
    dsync(dev)
    got = download(dsts[0].ptr, (H, W), dev)
    want = orc.unwarp_image_backward(*a2, order=3, mode="reflect", poly=orc.POLY_KERNEL)
    # (lines longer than one prefilter tile restart the recursion inside a halo: equal to the serial oracle up to the odd
    # float32 ulp)
    ok = np.count_nonzero(got != want) <= 32 and float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5
    # HBM bytes per frame of the path's launches (FETCH_SIZE x 2 + WRITE_SIZE, tools/pmc_spline.sh on the GPU box, committed under
    # profiles/): applies when the summary was taken on the kernels this run launched
    sp_traffic, sp_src = None, None
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "pmc_spline_latest.json")))
        names = sorted(n.split("<")[0] for n in j["kernels"])
        if all(n in k for n in names):
            sp_traffic = j["total_bytes_per_frame"]
            sp_src = "profiles/pmc_spline_latest.json: rocprofv3 PMC passes of %s (by construction %d B per pixel), not measured in this run" % (
                " + ".join(names), j["by_design_bytes_per_frame"] // (H * W))
        else:
            sp_src = "none: profiles/pmc_spline_latest.json is of %s" % " + ".join(names)
    except (OSError, ValueError, KeyError):
        pass
    out["cfg2_order3_cubic_spline"] = pf_entry(us, H * W, 8, k, ok, traffic=sp_traffic, traffic_source=sp_src,
                                            note="two launches per frame (prefilter of both axes, gather); pixels differing from the oracle by one float32 ulp: %d"
                                                 % int(np.count_nonzero(got != want)))

    # 16-bit detector frames (the element type tomography cameras deliver): the same kernel on narrower slab rows, scipy's
    # exact blend and integer store; 2 B read + 2 B written per pixel
    u16 = (img0 * 60000.0).astype(np.uint16)
    s16 = [F.DeviceBuffer(u16.nbytes, dev).upload(u16) for _ in range(4)]
    d16 = [F.DeviceBuffer(u16.nbytes, dev) for _ in range(4)]

    def radial_u16(i):
        F.check(L.dcp_unwarp_image_typed(s16[i % 4].ptr, d16[i % 4].ptr, F.DTYPE_BY_NAME["uint16"], H, W, W, 1, c2["xcenter"], c2["ycenter"],
                                         fa, nf, 1, 0, F.MEM_DEVICE, dev, dstream(i)))
    us = timed_launches(radial_u16, reps, dev, dispatch=True)
    k = F.last_kernel()
    radial_u16(0)
    dsync(dev)
    got = np.empty((H, W), np.uint16)
    F.check(L.dcp_memcpy(got.ctypes.data, d16[0].ptr, got.nbytes, F.COPY_D2H, dev, None))
    ok = np.array_equal(got, orc.unwarp_image_backward(u16, c2["xcenter"], c2["ycenter"], c2["list_fact"], poly=orc.POLY_KERNEL))
    out["cfg2_uint16_frame"] = pf_entry(us, H * W, 4, k, ok, note="uint16 in, uint16 out: 4 algorithmic bytes per pixel")
    for b in s16 + d16:
        b.free()

    def radial_default_stream(i):
        F.check(L.dcp_unwarp_image_f32(srcs[i % nring].ptr, dsts[i % nring].ptr, H, W, W, 1, c2["xcenter"], c2["ycenter"], fa, nf,
                                       1, 1, F.BLEND_F64LERP, F.MEM_DEVICE, dev, None))

    def two_pass(i):                                 # (frame i + 1 reads what frame i wrote: one stream, in order)
        radial_default_stream(i)                     # srcs[i] -> dsts[i]
        # dsts[i] -> srcs[i + 1]: the chain runs frame to frame through the ring
        F.check(L.dcp_perspective_image_f32(dsts[i % nring].ptr, srcs[(i + 1) % nring].ptr, H, W, W, 1, ca, 1, F.BLEND_F64LERP,
                                            F.MEM_DEVICE, dev, None))

    # (the two-pass chain overwrites ring inputs with corrected frames: it runs last among the 4096^2 cases, and frame 0 is
    # re-uploaded for its check)
    us = timed_launches(two_pass, reps // 2, dev)
    srcs[0].upload(img0)
    radial_default_stream(0)
    F.check(L.dcp_perspective_image_f32(dsts[0].ptr, dsts[1].ptr, H, W, W, 1, ca, 1, F.BLEND_F64LERP, F.MEM_DEVICE, dev, None))
    want = orc.correct_perspective_image(orc.unwarp_image_backward(*a2, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP), c3["list_coef"],
                                         blend=orc.BLEND_F64LERP)
    ok = np.array_equal(download(dsts[1].ptr, (H, W), dev), want)
    out["cfg3_two_pass_reference_semantics"] = entry(us, H * W, 16, "remap (radial) then remap (perspective): two launches", ok,
                                                     note="two resamplings: 16 algorithmic bytes per output pixel")
    return out


def config5(a, dev):
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    c5 = configs.cfg5()
    H, W = c5["shape"]
    fa, nf = F.fact_array(c5["list_fact"])
    img = np.random.default_rng(c5["seed"]).random((H, W), dtype=np.float32)
    nring = 4                                          # 4 x (268 + 268) MB = 2.1 GB
    src = [F.DeviceBuffer(img.nbytes, dev).upload(img) for _ in range(nring)]
    dst = [F.DeviceBuffer(img.nbytes, dev) for _ in range(nring)]

    def run(i):
        F.check(L.dcp_unwarp_image_f32(src[i % nring].ptr, dst[i % nring].ptr, H, W, W, 1, c5["xcenter"], c5["ycenter"], fa, nf, 1, 1,
                                       F.BLEND_F64LERP, dmem(), dev, dstream(i)))
    us = timed_launches(run, max(24, min(120, a.steps)), dev, dispatch=True)
    k = F.last_kernel()
    run(0)
    dsync(dev)
    ok = np.array_equal(download(dst[0].ptr, (H, W), dev),
                        orc.unwarp_image_backward(img, c5["xcenter"], c5["ycenter"], c5["list_fact"], poly=orc.POLY_KERNEL,
                                                  blend=orc.BLEND_F64LERP))
    for b in src + dst:
        b.free()
    return pf_entry(us, H * W, 8, k, ok, shape=[H, W], nfact=nf)


def color_frame(a, dev):
    """other_configs entry: util.unwarp_color_image_backward's kernel call on a device-resident 4096 x 4096 x 3 float32 image with
    config 2's calibration (the most-used entry of the fisheye examples) -- 12 B read + 12 B written per
    pixel, every channel checked against the oracle."""
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    c2 = configs.cfg2()
    H, W = c2["shape"]
    NC = 3
    fa, nf = F.fact_array(c2["list_fact"])
    rgb = np.random.default_rng(c2["seed"] + 77).random((H, W, NC), dtype=np.float32)
    nring = 6                                          # 6 x (201 + 201) MB = 2.4 GB: beyond the Infinity Cache
    src = [F.DeviceBuffer(rgb.nbytes, dev).upload(rgb) for _ in range(nring)]
    dst = [F.DeviceBuffer(rgb.nbytes, dev) for _ in range(nring)]

    def run(i):
        F.check(L.dcp_unwarp_color_image(src[i % nring].ptr, dst[i % nring].ptr, F.DTYPE_F32, H, W, NC, W * NC, NC, c2["xcenter"], c2["ycenter"],
                                         fa, nf, 1, F.BLEND_F64LERP, F.MEM_DEVICE, dev, dstream(i)))
    us = timed_launches(run, max(24, min(240, a.steps * 2)), dev, dispatch=True)
    k = F.last_kernel()
    run(0)
    dsync(dev)
    got = np.empty((H, W, NC), np.float32)
    F.check(L.dcp_memcpy(got.ctypes.data, dst[0].ptr, got.nbytes, F.COPY_D2H, dev, None))
    ok = all(np.array_equal(got[:, :, c], orc.unwarp_image_backward(np.ascontiguousarray(rgb[:, :, c]), c2["xcenter"], c2["ycenter"], c2["list_fact"],
                                                                    poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)) for c in range(NC))
    for b in src + dst:
        b.free()
    return pf_entry(us, H * W, 8 * NC, k, ok, shape=[H, W, NC],
                 note="interleaved float32 RGB, one coordinate and three blends per pixel; 24 algorithmic bytes per pixel")


# ----------------------------------------------------------------------------------------- config 4: the stack

class DevBlock:
    """A device allocation with a raw pointer: F.DeviceBuffer at N = 1, a torch tensor (needed by the collective) at N > 1."""

    def __init__(self, shape, dev, use_torch, device_str="cuda"):
        self.shape = tuple(int(s) for s in shape)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * 4
        if use_torch:
            import torch
            self.tensor = torch.empty(self.shape, dtype=torch.float32, device=device_str)
            self.ptr = self.tensor.data_ptr()
            self.buf = None
        else:
            self.tensor = None
            self.buf = F.DeviceBuffer(max(self.nbytes, 4), dev)
            self.ptr = self.buf.ptr

    def free(self):
        if self.buf is not None:
            self.buf.free()
        self.tensor = None


class SubBlock:
    """Projections [s0, s1) of a block (a view: same allocation)."""

    def __init__(self, block, s0, s1):
        per = int(np.prod(block.shape[1:], dtype=np.int64))
        self.shape = (s1 - s0,) + tuple(block.shape[1:])
        self.ptr = block.ptr + s0 * per * 4
        self.tensor = block.tensor[s0:s1] if getattr(block, "tensor", None) is not None else None


def fill_projections(block, dev, seed, upload=None):
    """Synthetic projections: one host chunk of noise replicated over the block (53.7 GB do not pass through PCIe)."""
    D, H, W = block.shape
    chunk = np.random.default_rng(seed).random((min(D, 8), H, W), dtype=np.float32)
    if upload is not None:
        upload(block, chunk)
        return chunk
    L = F.lib()
    F.check(L.dcp_memcpy(block.ptr, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, dev, None))
    done = chunk.shape[0]
    while done < D:                                    # doubling device-to-device copies
        n = min(done, D - done)
        F.check(L.dcp_memcpy(block.ptr + done * H * W * 4, block.ptr, n * H * W * 4, F.COPY_D2D, dev, None))
        done += n
    F.check(L.dcp_stream_synchronize(dev, None))
    return chunk


def hip_stack_launch(cfg, nrows, blend, dev, stream=None, row_start=0.0):
    """launch(vol_block, out_block, d_local): one dcp_unwarp_stack_rows_f32 call on device pointers."""
    L = F.lib()
    fa, nf = F.fact_array(cfg["list_fact"])
    _, H, W = cfg["shape"]

    def launch(vol, out, dl):
        F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr, dl, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                            float(row_start), nrows, 1, blend, F.MEM_DEVICE, dev, stream))
    launch.keep = (fa,)
    return launch


def stack_scaling(cfg, world, rank, dist, steps, warmup, nrows, make_block, launch, sync, fill, barrier_device="cuda",
                  verify=None, collective_name="all_gather_into_tensor", pipeline_blocks=4):
    """BASELINE config 4 with the projections sharded by depth over `world` ranks (discorpy_amd.stack.shard_bounds): time
    `steps` passes of (a) the local kernel alone and (b) the kernel followed by the all-gather that reassembles the
    (depth, nrows, width) block on every rank.  The max over ranks is what counts.  `make_block(shape)`, `launch(vol, out,
    d_local)`, `sync()` and `fill(block, seed)` abstract the device (HIP on the GPU boxes, the CPU oracle in the 2-rank gloo
    test of this very function: tests/test_bench_paths.py).  Returns the result dict on every rank."""
    from discorpy_amd import stack as st
    D, H, W = cfg["shape"]
    d0, d1 = st.shard_bounds(D, world, rank)
    dl = d1 - d0
    even = len({st.shard_bounds(D, world, r)[1] - st.shard_bounds(D, world, r)[0] for r in range(world)}) == 1
    vol = make_block((dl, H, W))
    out = make_block((dl, nrows, W))
    chunk = fill(vol, cfg["seed"] + rank)
    gather = world > 1 and even
    full = make_block((D, nrows, W)) if gather else None

    def timed(body):
        for _ in range(warmup):
            body()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            body()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        wall = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([wall], dtype=torch.float64, device=barrier_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall = float(tt[0])
        return wall * 1e3 / steps

    def compute_only():
        launch(vol, out, dl)

    def compute_and_gather():
        launch(vol, out, dl)
        dist.all_gather_into_tensor(full.tensor, out.tensor)

    nsub = min(pipeline_blocks, dl)

    def compute_and_gather_pipelined():
        # the shard in `nsub` depth sub-blocks: the (asynchronous) all-gather of sub-block s, every rank's piece landing in its place
        # of the depth-outer result, runs while the kernel of sub-block s + 1 computes (discorpy_amd.stack, pipeline=)
        pending = []
        for s_ in range(nsub):
            s0, s1 = st.shard_bounds(dl, nsub, s_)
            launch(SubBlock(vol, s0, s1), SubBlock(out, s0, s1), s1 - s0)
            pieces = [full.tensor[r * dl + s0:r * dl + s1] for r in range(world)]
            pending.append(dist.all_gather(pieces, out.tensor[s0:s1], async_op=True))
        for w in pending:
            w.wait()

    ms_compute = timed(compute_only)
    try:
        kernel = F.last_kernel()
    except Exception:      # noqa: BLE001 -- no library (CPU test of this function)
        kernel = ""
    ms_gather = timed(compute_and_gather) if gather else None
    verified = None
    if verify is not None:
        verified = bool(verify(chunk, out, full if gather else None, d0, dl))
    ms_pipe, verified_pipe = None, None
    if gather and nsub > 1:
        try:
            full.tensor.zero_()
            ms_pipe = timed(compute_and_gather_pipelined)
            if verify is not None:
                verified_pipe = bool(verify(chunk, out, full, d0, dl))
        except Exception as e:      # noqa: BLE001 -- a variant beside the plain all-gather: never fails the run
            ms_pipe, verified_pipe = None, "error: %r" % (e,)
    vox = float(D) * nrows * W
    res = {"what": "config 4: (%d, %d, %d) float32 stack, %d output rows of every projection, projections sharded by depth over "
                   "%d rank(s)" % (D, H, W, nrows, world),
           "depth_per_gpu": dl, "steps": steps,
           "compute_only": {"ms_per_step": round(ms_compute, 4), "Mpixels_per_s": round(vox / ms_compute / 1e3, 1),
                            "per_gpu_GBps": round(configs.BYTES_PER_PIXEL * dl * nrows * W / (ms_compute * 1e-3) / 1e9, 1)},
           "compute_plus_allgather": None if ms_gather is None else {
               "ms_per_step": round(ms_gather, 4), "Mpixels_per_s": round(vox / ms_gather / 1e3, 1),
               "gathered_bytes_received_per_gpu": int((D - dl) * nrows * W * 4), "collective": collective_name},
           "compute_plus_allgather_pipelined": None if ms_pipe is None else {
               "ms_per_step": round(ms_pipe, 4), "Mpixels_per_s": round(vox / ms_pipe / 1e3, 1), "depth_sub_blocks": nsub,
               "collective": "all_gather (list form, async_op) per sub-block under the next sub-block's kernel",
               "verified_vs_oracle": verified_pipe},
           "kernel": kernel, "verified_vs_oracle": verified}
    if world > 1 and not even:
        res["note"] = "depth does not divide evenly over the ranks: the timed all-gather needs even shards (discorpy_amd.stack pads ragged ones)"
    if world == 1:
        res["note"] = "one rank: the local block IS the whole result, no collective"
    for b in (vol, out, full):
        if b is not None:
            b.free()
    return res


def hip_stack_verify(cfg, nrows, dev, a, orc_blend_name="f64lerp"):
    """verify(chunk, out_block, full_block, d0, dl): projections 0 and (dl - 1) of the local result (and this rank's first
    projection inside the gathered block) against the oracle -- every row, all of the width."""
    def verify(chunk, out, full, d0, dl):
        orc = oracle_module(a.cpu_threads)
        _, H, W = cfg["shape"]
        picks = sorted({0, dl - 1})
        ok = True
        for d in picks:
            src = chunk[d % chunk.shape[0]][None]          # the block is the chunk replicated
            want = orc.unwarp_stack_rows(src, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], 0, nrows, coord_round_f32=True,
                                         poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
            got = download(out.ptr, (1, nrows, W), dev, offset=d * nrows * W * 4)
            ok = ok and np.array_equal(got, want)
            if full is not None:
                got = download(full.ptr, (1, nrows, W), dev, offset=(d0 + d) * nrows * W * 4)
                ok = ok and np.array_equal(got, want)
        return ok
    return verify


def stack_on_hip(a, world, rank, dev, dist, backend, depth, nrows, steps, warmup, blend):
    """stack_scaling on this rank's GPU."""
    L = F.lib()
    cfg = configs.cfg4(depth)
    use_torch = world > 1
    stream = None
    if use_torch:
        import torch
        stream = torch.cuda.current_stream().cuda_stream

    def sync():
        F.check(L.dcp_stream_synchronize(dev, None))
        if use_torch:
            import torch
            torch.cuda.synchronize()
    return stack_scaling(cfg, world, rank, dist, steps, warmup, nrows,
                         make_block=lambda shape: DevBlock(shape, dev, use_torch),
                         launch=hip_stack_launch(cfg, nrows, blend, dev, stream), sync=sync,
                         fill=lambda block, seed: fill_projections(block, dev, seed),
                         barrier_device="cuda" if backend == "nccl" else "cpu",
                         verify=hip_stack_verify(cfg, nrows, dev, a),
                         collective_name="all_gather_into_tensor, backend nccl = RCCL over xGMI" if backend == "nccl"
                         else "all_gather_into_tensor, backend %s (test hook DCP_BENCH_BACKEND: not RCCL)" % backend)


def free_device_bytes(dev):
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        fr, tot = C.c_size_t(0), C.c_size_t(0)
        if hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0:
            return int(fr.value)
    except Exception:      # noqa: BLE001
        pass
    return None


def stack_typed_shard(a, dev, dtype="uint16"):
    """One 8-GPU shard of config 4 as projections of another element type (64 of them, every row): dcp_unwarp_stack_rows_typed.
    uint16 is what tomography detectors deliver; int32 and float64 are the other types on stack_wg_kernel."""
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    cfg = configs.cfg4(64)
    D, H, W = cfg["shape"]
    fa, nf = F.fact_array(cfg["list_fact"])
    dt = np.dtype(dtype)
    es = dt.itemsize
    chunk = (np.random.default_rng(cfg["seed"] + 5).random((4, H, W), dtype=np.float32) * 60000.0).astype(dt)
    vol = F.DeviceBuffer(D * H * W * es, dev)
    out = F.DeviceBuffer(D * H * W * es, dev)
    for d in range(0, D, 4):
        F.check(L.dcp_memcpy(vol.ptr + d * H * W * es, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, dev, None))
    code = F.DTYPE_BY_NAME[dt.name]

    def run(_i):
        F.check(L.dcp_unwarp_stack_rows_typed(vol.ptr, out.ptr, code, 0, D, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf, 0.0, H, 1,
                                              F.MEM_DEVICE, dev, None))
    us = timed_launches(run, 12, dev, settle_ms=60.0)
    k = F.last_kernel()
    got = np.empty((1, H, W), dt)
    F.check(L.dcp_memcpy(got.ctypes.data, out.ptr + 3 * H * W * es, got.nbytes, F.COPY_D2H, dev, None))
    want = orc.unwarp_chunk_slices_backward(chunk[3:4], cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], 0, H - 1, poly=orc.POLY_KERNEL)
    ok = np.array_equal(got, want)
    vol.free()
    out.free()
    return entry(us, D * H * W, 2 * es, k, ok, shape=[D, H, W], note="%s projections, every row: %d algorithmic bytes per voxel" % (dt.name, 2 * es))


def uint16_frames_batched(a, dev, n=16):
    """`n` 4096 x 4096 uint16 detector frames of ONE calibration in one (n, H, W) device array and ONE call -- what
    post.unwarp_images_backward does with such an array: dcp_unwarp_stack_rows_typed with coord_round_f32 = 2
    (unwarp_image_backward's semantics: whole-frame clip, no row band), the stack kernel of that element type; the coordinate work of
    a pixel position is shared by the n frames.  Frame 3 against the oracle's unwarp_image_backward."""
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    c2 = configs.cfg2()
    H, W = c2["shape"]
    fa, nf = F.fact_array(c2["list_fact"])
    chunk = (np.random.default_rng(c2["seed"] + 9).random((4, H, W), dtype=np.float32) * 60000.0).astype(np.uint16)
    vol = F.DeviceBuffer(n * H * W * 2, dev)
    out = F.DeviceBuffer(n * H * W * 2, dev)
    for d in range(0, n, 4):
        F.check(L.dcp_memcpy(vol.ptr + d * H * W * 2, chunk.ctypes.data, chunk.nbytes, F.COPY_H2D, dev, None))
    code = F.DTYPE_BY_NAME["uint16"]

    def run(_i):
        F.check(L.dcp_unwarp_stack_rows_typed(vol.ptr, out.ptr, code, 0, n, H, W, H * W, W, c2["xcenter"], c2["ycenter"], fa, nf, 0.0, H, 2,
                                              F.MEM_DEVICE, dev, None))
    us = timed_launches(run, 24, dev, settle_ms=60.0) / n
    k = F.last_kernel()
    got = np.empty((H, W), np.uint16)
    F.check(L.dcp_memcpy(got.ctypes.data, out.ptr + 3 * H * W * 2, got.nbytes, F.COPY_D2H, dev, None))
    ok = np.array_equal(got, orc.unwarp_image_backward(chunk[3], c2["xcenter"], c2["ycenter"], c2["list_fact"], poly=orc.POLY_KERNEL))
    vol.free()
    out.free()
    return entry(us, H * W, 4, k, ok, frames_per_call=n, note="per frame of %d uint16 frames of one calibration in one call (a 3-D device array through "
                 "unwarp_images_backward): 4 algorithmic bytes per pixel" % n)


def stack_one_gpu_cases(a, dev):
    """other_configs entries of config 4 on one GPU: one sinogram of a depth-256 shard (unwarp_slice_backward, float64
    coordinates, launch-bound) -- the whole-stack number is stack_scaling's compute_only at N = 1."""
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    cfg = configs.cfg4(256)
    D, H, W = cfg["shape"]
    fa, nf = F.fact_array(cfg["list_fact"])
    vol = DevBlock((D, H, W), dev, False)
    out = DevBlock((D, 1, W), dev, False)
    chunk = fill_projections(vol, dev, cfg["seed"])
    row = 1277.0

    def run(i):
        F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr, D, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf, row, 1, 0,
                                            F.BLEND_F64LERP, F.MEM_DEVICE, dev, None))
    us = timed_launches(run, 200, dev, settle_ms=50.0)
    k = F.last_kernel()
    got = download(out.ptr, (chunk.shape[0], 1, W), dev)
    want = orc.unwarp_stack_rows(chunk, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], row, 1, coord_round_f32=False,
                                 poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
    ok = np.array_equal(got, want)
    vol.free()
    out.free()
    return entry(us, D * W, 12, k, ok, note="unwarp_slice_backward of a depth-256 shard: one output row per projection reads two "
                                            "source rows (12 B per voxel); 2.6 MB per launch -- launch-bound")


# ----------------------------------------------------------------------------------------- exchange without torch (C ABI only)

def rccl_comm_report(comm, world):
    """What the communicator of the C-ABI exchange says about ITSELF (dcp_rccl_comm_info: ncclCommCount / ncclCommUserRank /
    ncclCommCuDevice / ncclGetVersion, the librccl file the symbols were bound from, the shard depths of the last agreement)."""
    import ctypes as C
    L = F.lib()
    info = (C.c_int64 * 10)()
    depths = (C.c_int64 * max(world, 1))()
    path = C.create_string_buffer(1024)
    F.check(L.dcp_rccl_comm_info(comm, info, 10, depths, world, path, 1024))
    lib = path.value.decode(errors="replace")
    v = int(info[3])
    return {"nccl_comm_count": int(info[0]), "nccl_comm_user_rank": int(info[1]), "nccl_comm_device": int(info[2]),
            "nccl_version_code": v, "nccl_version": ("%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100)) if v >= 10000 else str(v),
            "world_given": int(info[4]), "rank_given": int(info[5]), "hip_device": int(info[6]), "shard_agreement_in_force": bool(info[7]),
            "exchanges": int(info[8]), "agreements": int(info[9]), "shard_depths": [int(d) for d in depths[:world]], "librccl": lib,
            # tests/c/libfake_rccl.so (DCP_RCCL_PATH on the one-GPU test boxes) reports version 1: never mistaken for RCCL
            "rccl_is_stand_in": bool(v < 10000 or "fake_rccl" in lib)}


def rccl_debug_env(env, tag):
    """NCCL_DEBUG=INFO into a side file per process under gpurun_out/rccl_debug/ (unless the caller already set NCCL_DEBUG): the
    topology, algorithm and protocol lines of the first multi-GPU run, kept beside the numbers."""
    if "NCCL_DEBUG" in env and env.get("DCP_BENCH_NCCL_DEBUG_IS_OURS") != "1":
        return None
    d = os.path.join(ROOT, "gpurun_out", "rccl_debug")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return None
    env["NCCL_DEBUG"] = "INFO"
    env["DCP_BENCH_NCCL_DEBUG_IS_OURS"] = "1"
    env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,ENV,TUNING")
    env["NCCL_DEBUG_FILE"] = os.path.join(d, tag + ".%h.%p.log")
    return d


def rccl_debug_digest(d, tag, limit=16):
    """A few distinct lines of the side files written under rccl_debug_env (ranks, transport, algorithm / protocol choices)."""
    import glob
    import re
    if not d:
        return None
    files = sorted(glob.glob(os.path.join(d, tag + ".*.log")))
    keep, seen = [], set()
    pat = re.compile(r"nranks|nRanks|Algo|Proto|via P2P|via SHM|via NET|Connected all|Using network|RCCL version|NCCL version|comm 0x", re.I)
    for f in files:
        try:
            for ln in open(f, errors="replace"):
                if "NCCL" in ln and pat.search(ln):
                    key = re.sub(r"0x[0-9a-f]+|\d+", "#", ln.split("NCCL", 1)[1])[:80]
                    if key not in seen and len(keep) < limit:
                        seen.add(key)
                        keep.append(ln.strip()[:220])
        except OSError:
            pass
    return {"dir": os.path.relpath(d, ROOT), "files": len(files), "sample_lines": keep}


def torch_rccl_report(dist, backend, world, rank, dev_index, dbg_dir):
    """COLLECTIVE (every rank calls it).  What the torch.distributed job that carried stack_scaling's all-gather consisted of, from
    the collectives themselves: an all-reduce of ones counts the ranks that took part, an object all-gather lists their devices."""
    import torch
    t = torch.ones(1, dtype=torch.float32, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    ranks = [None] * world
    dist.all_gather_object(ranks, {"rank": rank, "device": dev_index, "pid": os.getpid()})
    ver, lib = None, None
    if backend == "nccl":
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001
            ver = None
    try:
        for ln in open("/proc/self/maps"):
            if "librccl" in ln or "libnccl" in ln:
                lib = ln.split()[-1]
                break
    except OSError:
        pass
    rep = {"backend": backend, "ranks_in_all_reduce": int(round(float(t.item()))), "torch_world_size": dist.get_world_size(), "n_gpus": world,
           "nccl_version": ver, "librccl": lib, "ranks": ranks,
           # DCP_BENCH_BACKEND=gloo (ranks sharing the one GPU of a test box): not an RCCL number, and the line says so
           "rccl_is_stand_in": backend != "nccl"}
    if rank == 0:
        rep["debug_log"] = rccl_debug_digest(dbg_dir, "torch")
    if rep["ranks_in_all_reduce"] != world or len({r["rank"] for r in ranks if r}) != world:
        rep["error"] = "the collective saw %d ranks, the launcher started %d" % (rep["ranks_in_all_reduce"], world)
    return rep


def native_rccl_child(a):
    """One rank of config 4 with the exchange done by dcp_unwarp_stack_rows_rccl_f32 (dlopen'ed librccl: ncclAllGather in place on
    the kernel's stream; pipelined: grouped ncclBroadcasts of depth sub-blocks on a side stream) -- no torch in this process.
    The ncclUniqueId travels through --idfile (rank 0 writes it, the others poll)."""
    import ctypes as C
    L = F.lib()
    F.require_device()
    world, rank = a.child_world, a.child_rank
    dev = int(os.environ.get("DCP_BENCH_DEVICE", rank))
    idbuf = (C.c_char * 128)()
    if rank == 0:
        F.check(L.dcp_rccl_unique_id(idbuf, 128))
        tmp = a.idfile + ".tmp"
        open(tmp, "wb").write(bytes(idbuf))
        os.replace(tmp, a.idfile)
    else:
        t0 = time.perf_counter()
        while not os.path.exists(a.idfile):
            if time.perf_counter() - t0 > 90:
                raise RuntimeError("no RCCL unique id from rank 0 after 90 s")
            time.sleep(0.05)
        idbuf = (C.c_char * 128).from_buffer_copy(open(a.idfile, "rb").read())
    comm = C.c_void_p()
    F.check(L.dcp_rccl_comm_create(C.byref(comm), world, rank, idbuf, dev))
    cfg = configs.cfg4(a.depth)
    D, H, W = cfg["shape"]
    if D % world:
        raise RuntimeError("depth %d does not divide over %d ranks" % (D, world))
    dl, nrows = D // world, a.rows
    vol = DevBlock((dl, H, W), dev, False)
    full = DevBlock((D, nrows, W), dev, False)
    chunk = fill_projections(vol, dev, cfg["seed"] + rank)
    fa, nf = F.fact_array(cfg["list_fact"])
    res = {"rank": rank, "world": world, "device": dev, "depth_per_gpu": dl}

    def run(pipeline):
        F.check(L.dcp_unwarp_stack_rows_rccl_f32(vol.ptr, full.ptr, dl, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf, 0.0, nrows, 1,
                                                 F.BLEND_F64LERP, comm, pipeline, None))

    def check():
        orc = oracle_module(a.cpu_threads)
        ok = True
        for r, d in ((rank, 0), ((rank + 1) % world, dl - 1)):        # one projection of this rank's block, one of a neighbour's
            src = chunk if r == rank else np.random.default_rng(cfg["seed"] + r).random((min(dl, 8), H, W), dtype=np.float32)
            want = orc.unwarp_stack_rows(src[d % src.shape[0]][None], cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], 0, nrows,
                                         coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
            got = download(full.ptr, (1, nrows, W), dev, offset=(r * dl + d) * nrows * W * 4)
            ok = ok and bool(np.array_equal(got, want))
        return ok
    zeros = np.zeros((nrows, W), np.float32)
    for name, pipeline in (("allgather", 1), ("allgather_pipelined", 4)):
        for r, d in ((rank, 0), ((rank + 1) % world, dl - 1)):      # the projections check() reads must come from THIS variant
            F.check(L.dcp_memcpy(full.ptr + (r * dl + d) * nrows * W * 4, zeros.ctypes.data, zeros.nbytes, F.COPY_H2D, dev, None))
        F.check(L.dcp_rccl_comm_fixed_shards(comm, 0))
        run(pipeline)                                   # warm-up; its shard agreement (a collective + a host wait) also lines the ranks up
        F.check(L.dcp_stream_synchronize(dev, None))
        # the timed calls repeat the agreed shapes on every rank: say so, and they are stream-ordered end to end (no agreement, no host
        # wait between two exchanges -- ADVICE r4: with it, back-to-back calls were synchronous)
        F.check(L.dcp_rccl_comm_fixed_shards(comm, 1))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run(pipeline)
        F.check(L.dcp_stream_synchronize(dev, None))
        res[name + "_ms"] = (time.perf_counter() - t0) * 1e3 / a.steps
        res[name + "_verified"] = check()
    res["rccl"] = rccl_comm_report(comm, world)
    res["bytes_received"] = int((D - dl) * nrows * W * 4)
    F.check(L.dcp_rccl_comm_destroy(comm))
    vol.free()
    full.free()
    print(json.dumps(res), flush=True)


def native_peer_child(a):
    """Config 4 with every GPU driven from THIS one process and the exchange done by peer copies (dcp_unwarp_stack_rows_peer_f32:
    hipMemcpyPeerAsync pushes, one per xGMI link and device) -- no torch, no RCCL."""
    import ctypes as C
    L = F.lib()
    F.require_device()
    world = a.child_world
    cfg = configs.cfg4(a.depth)
    D, H, W = cfg["shape"]
    if D % world:
        raise RuntimeError("depth %d does not divide over %d devices" % (D, world))
    dl, nrows = D // world, a.rows
    share = os.environ.get("DCP_BENCH_DEVICE")              # test hook: every slot on one GPU
    devices = [int(share) if share is not None else g for g in range(world)]
    vols = [DevBlock((dl, H, W), d, False) for d in devices]
    outs = [DevBlock((D, nrows, W), d, False) for d in devices]
    chunks = [fill_projections(v, d, cfg["seed"] + g) for g, (v, d) in enumerate(zip(vols, devices))]
    fa, nf = F.fact_array(cfg["list_fact"])
    vp = (C.c_void_p * world)(*[v.ptr for v in vols])
    op = (C.c_void_p * world)(*[o.ptr for o in outs])
    da = (C.c_int * world)(*devices)

    def run():
        F.check(L.dcp_unwarp_stack_rows_peer_f32(vp, op, D, H, W, H * W, W, cfg["xcenter"], cfg["ycenter"], fa, nf, 0.0, nrows, 1, F.BLEND_F64LERP,
                                                 da, world, 1))
    run()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()                                           # (returns after every stream has drained)
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    orc = oracle_module(a.cpu_threads)
    ok = True
    for g, r in ((0, 0), (0, world - 1), (world - 1, 0)):   # slot g's result, the block that slot r computed
        want = orc.unwarp_stack_rows(chunks[r][0][None], cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], 0, nrows, coord_round_f32=True,
                                     poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
        got = download(outs[g].ptr, (1, nrows, W), devices[g], offset=r * dl * nrows * W * 4)
        ok = ok and bool(np.array_equal(got, want))
    for b in vols + outs:
        b.free()
    print(json.dumps({"world": world, "devices": devices, "peer_copies_ms": ms, "verified": ok}), flush=True)


def native_exchange_variants(a, world, depth, steps=3, timeout=None):
    """Rank 0 of the torch job calls this after stack_scaling (the other ranks wait at a barrier, their blocks freed): the same
    workload with the exchange done WITHOUT torch -- (a) one C-ABI process per GPU and a native RCCL all-gather, plain and
    pipelined, (b) one process driving every GPU with peer copies.  Child processes with a timeout each: a hang or a crash in a
    path that no single-GPU box can exercise at world > 1 must not cost the run its JSON line."""
    import subprocess
    import tempfile
    if timeout is None:
        timeout = 150 + 20 * world
    D = depth
    vox = float(D) * 2560 * 2560
    out = {}
    base = [sys.executable, os.path.abspath(__file__), "--depth", str(D), "--rows", "2560", "--steps", str(steps), "--child-world", str(world)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dbg_dir = rccl_debug_env(env, "native")

    def last_json(text):
        lines = [ln for ln in (text or "").strip().split("\n") if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    # (a) native RCCL, one process per GPU
    try:
        with tempfile.TemporaryDirectory() as td:
            idfile = os.path.join(td, "rccl_id")
            procs = [subprocess.Popen(base + ["--native-child", "rccl", "--child-rank", str(r), "--idfile", idfile], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT) for r in range(world)]
            t_end = time.perf_counter() + timeout
            res, errs = [], []
            for p_ in procs:
                try:
                    so, se = p_.communicate(timeout=max(1.0, t_end - time.perf_counter()))
                    j = last_json(so) if p_.returncode == 0 else None
                    (res if j else errs).append(j or (se or so)[-300:])
                except subprocess.TimeoutExpired:
                    p_.kill()
                    p_.communicate()
                    errs.append("timeout after %d s" % timeout)
        seen = sorted({r["rccl"]["nccl_comm_count"] for r in res}) if res and all("rccl" in r for r in res) else None
        if errs or len(res) != world:
            out["native_rccl"] = {"error": errs[:2]}
        elif seen != [world]:
            # the entry fails, not the bench: the communicators did not span the ranks the launcher started
            out["native_rccl"] = {"error": "ncclCommCount reports %r ranks, %d processes were launched" % (seen, world),
                                  "per_rank": [r.get("rccl") for r in res]}
        else:
            r0 = sorted(res, key=lambda r: r["rank"])
            out["native_rccl_comm"] = {
                "ranks_seen_by_rccl": seen[0], "n_gpus": world, "nccl_version": r0[0]["rccl"]["nccl_version"], "librccl": r0[0]["rccl"]["librccl"],
                "rccl_is_stand_in": any(r["rccl"]["rccl_is_stand_in"] for r in r0),
                "user_ranks": [r["rccl"]["nccl_comm_user_rank"] for r in r0], "devices": [r["rccl"]["nccl_comm_device"] for r in r0],
                "shard_depths": r0[0]["rccl"]["shard_depths"], "bytes_received_per_rank": [r["bytes_received"] for r in r0],
                "exchanges_per_rank": [r["rccl"]["exchanges"] for r in r0], "agreements_per_rank": [r["rccl"]["agreements"] for r in r0],
                "debug_log": rccl_debug_digest(dbg_dir, "native")}
            for name in ("allgather", "allgather_pipelined"):
                ms = max(r[name + "_ms"] for r in res)
                out["native_rccl_" + name] = {"ms_per_step": round(ms, 4), "Mpixels_per_s": round(vox / ms / 1e3, 1),
                                              "verified_vs_oracle": all(r[name + "_verified"] for r in res),
                                              "librccl": os.environ.get("DCP_RCCL_PATH") or "the one next to the HIP runtime",
                                              "how": "dcp_unwarp_stack_rows_rccl_f32, one C-ABI process per GPU (no torch), max over ranks of "
                                                     "wall time per step incl. the kernel"}
    except Exception as e:      # noqa: BLE001
        out["native_rccl"] = {"error": repr(e)}
    # (b) peer copies, one process for all GPUs
    try:
        r = subprocess.run(base + ["--native-child", "peer"], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        j = last_json(r.stdout) if r.returncode == 0 else None
        if j:
            out["peer_copies"] = {"ms_per_step": round(j["peer_copies_ms"], 4), "Mpixels_per_s": round(vox / j["peer_copies_ms"] / 1e3, 1),
                                  "verified_vs_oracle": j["verified"],
                                  "how": "dcp_unwarp_stack_rows_peer_f32: one process drives every GPU, hipMemcpyPeerAsync pushes; wall time per "
                                         "call incl. stream creation and the final synchronisation"}
        else:
            out["peer_copies"] = {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:      # noqa: BLE001
        out["peer_copies"] = {"error": repr(e)}
    return out


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: re-launch as N ranks through torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and pass rank 0's JSON line through."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    args = [x for x in (sys.argv[1:] if argv is None else list(argv))]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env, cwd=os.getcwd()).returncode


def grid_search_centres(a, dev):
    """other_configs entry: the grid search of examples/example_05.py:62-65 -- unwarp_slice_backward for 11 x 11 candidate centres
    on a device-resident depth-256 shard of config 4 -- as ONE dcp_unwarp_stack_rows_centres_f32 call, against 121 single calls."""
    import ctypes as C
    L = F.lib()
    orc = oracle_module(a.cpu_threads)
    cfg = configs.cfg4(256)
    D, H, W = cfg["shape"]
    fa, nf = F.fact_array(cfg["list_fact"])
    vol = DevBlock((D, H, W), dev, False)
    chunk = fill_projections(vol, dev, cfg["seed"])
    cents = [(cfg["xcenter"] + dx, cfg["ycenter"] + dy) for dx in range(-100, 120, 20) for dy in range(-100, 120, 20)]
    K = len(cents)
    out = DevBlock((K, D, 1, W), dev, False)
    xa, ya = (C.c_double * K)(*[c[0] for c in cents]), (C.c_double * K)(*[c[1] for c in cents])
    row = 1277.0

    def batched(_i):
        F.check(L.dcp_unwarp_stack_rows_centres_f32(vol.ptr, out.ptr, D, H, W, H * W, W, xa, ya, K, fa, nf, row, 1, 0, F.BLEND_F64LERP, F.MEM_DEVICE,
                                                    dev, None))

    def single(i):
        k = i % K
        F.check(L.dcp_unwarp_stack_rows_f32(vol.ptr, out.ptr + k * D * W * 4, D, H, W, H * W, W, cents[k][0], cents[k][1], fa, nf, row, 1, 0,
                                            F.BLEND_F64LERP, F.MEM_DEVICE, dev, None))
    us = timed_launches(batched, 40, dev, settle_ms=60.0)
    kernel = F.last_kernel()
    us_single = timed_launches(single, 2 * K, dev, settle_ms=60.0) * K
    batched(0)
    ok = True
    for k in (0, K // 2, K - 1):
        want = orc.unwarp_stack_rows(chunk, cents[k][0], cents[k][1], cfg["list_fact"], row, 1, coord_round_f32=False, poly=orc.POLY_KERNEL,
                                     blend=orc.BLEND_F64LERP)
        got = download(out.ptr, (chunk.shape[0], 1, W), dev, offset=k * D * W * 4)
        ok = ok and bool(np.array_equal(got, want))
    # algorithmic bytes of the BATCHED call: every voxel of the (K, D, 1, W) result written once, and the source rows any centre reads
    # (their union over the K centres: dcp_stack_row_band per centre) read once per projection -- not K times the single call's 12 B
    lo, hi = C.c_int64(0), C.c_int64(0)
    b0, b1 = H, 0
    for cx, cy in cents:
        F.check(L.dcp_stack_row_band(H, W, cx, cy, fa, nf, row, 1, C.byref(lo), C.byref(hi)))
        b0, b1 = min(b0, lo.value), max(b1, lo.value + hi.value)          # (band_start, band_rows)
    alg = 4.0 * K * D * W + 4.0 * D * max(b1 - b0, 1) * W
    vol.free()
    out.free()
    return entry(us, K * D * W, round(alg / (K * D * W), 4), kernel, ok, centres=K, depth=D,
                 source_rows_read_per_projection=int(b1 - b0), algorithmic_bytes_per_launch=int(alg),
                 the_same_as_single_calls_us=round(us_single, 2),
                 note="%d candidate centres x one sinogram of a depth-%d shard in one launch.  Algorithmic bytes of THIS call: the (K, D, 1, W) "
                      "result written once + the union of the source rows the centres read, once per projection (a single call moves 12 B per "
                      "voxel; here the source is shared by all centres)" % (K, D))


# ----------------------------------------------------------------------------------------- main

def init_dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (single-GPU boxes): DCP_BENCH_BACKEND=gloo + DCP_BENCH_DEVICE=0 let two ranks share one GPU
    backend = os.environ.get("DCP_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("DCP_BENCH_DEVICE", local_rank))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    return world, rank, dev_index, dist, backend


def stack_traffic(a, world, kernel):
    """roofline.traffic of the stack workload: the PMC figure committed under profiles/ applies to a 256-projection shard, every row."""
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        j = json.load(open(pmc)).get("stack_shard256")
        if j and a.depth // world == 256 and a.rows == 2560 and kernel.split("<")[0] in j.get("rocprof_kernel_name", ""):
            return {"traffic": j.get("hbm_bytes_per_launch"), "traffic_stale": json.load(open(pmc)).get("kernel_sources_sha256") != kernel_sources_digest(),
                    "traffic_source": "profiles/pmc_latest.json: rocprofv3 PMC passes of %s on a 256-projection shard, not measured in this run" % j.get("rocprof_kernel_name")}
    except Exception:      # noqa: BLE001
        pass
    return {"traffic": None, "traffic_source": None, "traffic_stale": None}


def stack_main(a, world, rank, dev, dist, backend):
    """--workload stack: config 4 IS the metric (strong scaling: the stack is fixed)."""
    blend = BLEND_NAMES[a.blend]
    if a.shard == "rows":
        # no collective: every rank holds the whole stack and owns output rows [r0, r1) of every projection -- complete
        # sinograms for its rows, which is what a per-sinogram reconstructor downstream wants
        from discorpy_amd import stack as st
        L = F.lib()
        cfg = configs.cfg4(a.depth)
        D, H, W = cfg["shape"]
        r0, r1 = st.row_shard_bounds(a.rows, world, rank)
        vol = DevBlock((D, H, W), dev, False)
        out = DevBlock((D, max(r1 - r0, 1), W), dev, False)
        fill_projections(vol, dev, cfg["seed"])            # the same stack on every rank
        launch = hip_stack_launch(cfg, r1 - r0, blend, dev, None, row_start=float(r0))

        def sync():
            F.check(L.dcp_stream_synchronize(dev, None))
        for _ in range(a.warmup):
            launch(vol, out, D)
        sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            launch(vol, out, D)
        sync()
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall = float(tt[0])
        ms = wall * 1e3 / a.steps
        res = {"ms_per_step": ms, "Mpixels_per_s": float(D) * a.rows * W / ms / 1e3, "rows_per_gpu": r1 - r0, "kernel": F.last_kernel()}
        mode = "row-sharded, no collective (every rank holds the stack, owns output rows of every projection)"
    else:
        r = stack_on_hip(a, world, rank, dev, dist, backend, a.depth, a.rows, a.steps, a.warmup, blend)
        use = r["compute_plus_allgather"] if (r["compute_plus_allgather"] and not a.no_gather) else r["compute_only"]
        res = {"ms_per_step": use["ms_per_step"], "Mpixels_per_s": use["Mpixels_per_s"], "kernel": r["kernel"], "detail": r}
        mode = "depth-sharded, %s" % ("RCCL all-gather of the (depth, rows, W) block" if use is r["compute_plus_allgather"] else "no collective")
    if rank == 0:
        cfg = configs.cfg4(a.depth)
        D, H, W = cfg["shape"]
        per_gpu = configs.BYTES_PER_PIXEL * float(D) * a.rows * W / world
        achieved = per_gpu / (res["ms_per_step"] * 1e-3) / 1e9
        print(json.dumps({
            "metric": "Mpixels/s unwarp of a (depth, 2560, 2560) stack, rows of every projection", "value": round(res["Mpixels_per_s"], 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(res["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (uniform [0,1) float32 projections, device-resident)",
            "config": {"workload": cfg["name"], "depth": D, "rows": a.rows, "width": W, "blend": a.blend, "parallelism": mode},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": configs.HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / configs.HBM_PEAK_GBPS, 4), "kernel": res["kernel"], **stack_traffic(a, world, res["kernel"])},
            "detail": res.get("detail")}), flush=True)


def main(argv=None):
    a = parse(argv)
    if a.e2e_child:
        e2e_child()
        return
    if a.native_child == "rccl":
        native_rccl_child(a)
        return
    if a.native_child == "peer":
        native_peer_child(a)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.no_spawn:
        sys.exit(spawn_ranks(a, argv))
    # the drop-in caller's number (NumPy in -> NumPy out) is measured FIRST, in child processes, before this process opens the
    # GPU: with a second process holding a context on the device the /opt/rocm runtime's two-directional host path ran at half
    # its speed (3.76 ms against 1.83 ms per 4096^2 frame)
    e2e = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not a.no_extras and a.workload == "frame":
        try:
            e2e = end_to_end_numpy()
        except Exception as e:      # noqa: BLE001 -- context only
            e2e = {"error": repr(e)}
    torch_dbg_dir = rccl_debug_env(os.environ, "torch") if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not a.no_extras else None
    world, rank, dev_index, dist, backend = init_dist()
    n_gpus = world
    if a.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)

    L = F.lib()
    F.require_device()
    for kv in a.option:
        k, v = kv.split("=")
        F.set_option(k, int(v))
    dev = dev_index if world > 1 else -1
    if a.workload == "stack":
        stack_main(a, world, rank, dev, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        return
    cfg = configs.cfg2()
    cfg["order"] = a.order
    H, W = cfg["shape"]
    blend = BLEND_NAMES[a.blend]
    rng = np.random.default_rng(cfg["seed"] + 1000 * rank)
    fa, nf = F.fact_array(cfg["list_fact"])

    # device-resident batch: `batch` distinct input frames + as many output frames
    # (ONE allocation per side, the frames at a constant pitch: the ring is a (batch, H, W) array -- what a caller holding a
    # stack of frames of one camera has, and what the batch entry point recognises as such)
    frame_bytes = H * W * 4
    ring_src, ring_dst = F.DeviceBuffer(frame_bytes * a.batch, dev), F.DeviceBuffer(frame_bytes * a.batch, dev)
    srcs, dsts, img0 = [], [], None
    for i in range(a.batch):
        img = rng.random((H, W), dtype=np.float32)
        if i == 0:
            img0 = img
        srcs.append(RingFrame(ring_src, i * frame_bytes).upload(img))
        dsts.append(RingFrame(ring_dst, i * frame_bytes))

    # how the launches leave the host (--dispatch): the frames are independent -- one calibration per call, one call per frame
    set_dispatch(a.dispatch, dev)
    lstreams = _DISPATCH["streams"]
    frame_mem = dmem()
    launch_no = [0]

    def ring_pass():
        for s, d in zip(srcs, dsts):
            st = lstreams[launch_no[0] & 1].ptr if lstreams else None
            launch_no[0] += 1
            rc = L.dcp_unwarp_image_f32(s.ptr, d.ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                        a.order, 1, blend, frame_mem, dev, st)
            if rc:
                F.check(rc)

    def sync():
        for st in lstreams:
            st.synchronize()
        F.check(L.dcp_stream_synchronize(dev, None))
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    # device time of a region that runs on BOTH streams: the start event is recorded on the first and the second waits for it (nothing
    # of the region starts before it); at the end the first stream waits for an event of the second before the stop event is
    # recorded on it -- elapsed(start, stop) covers every launch of the region on either stream, gaps included
    join_ev = [F.Event(dev) for _ in range(4)] if lstreams else []

    def mark_begin(e):
        if lstreams:
            e.record(lstreams[0].ptr)
            lstreams[1].wait_event(e)
        else:
            e.record()

    def mark_end(e, k=0):
        if lstreams:
            join_ev[k].record(lstreams[1].ptr)
            lstreams[0].wait_event(join_ev[k])
            e.record(lstreams[0].ptr)
        else:
            e.record()

    # let the clocks reach their sustained level before anything is counted: ring passes for at least --settle-ms, then on until three
    # consecutive passes agree to 1.5 % (device time per pass; at most 8 x --settle-ms) -- a cold box ramps for longer than a warm one,
    # and two consecutive runs of the driver's command differed by 3.6 % with the fixed 250 ms
    settle = {"ms": 0.0, "passes": 0, "last_pass_ms": []}
    if a.settle_ms > 0:
        t_settle = time.perf_counter()
        s0, s1 = F.Event(dev), F.Event(dev)
        recent = []
        while True:
            mark_begin(s0)
            ring_pass()
            mark_end(s1, 1)
            s1.synchronize()
            recent = (recent + [s0.elapsed_ms(s1)])[-3:]
            settle["passes"] += 1
            el = (time.perf_counter() - t_settle) * 1e3
            if el >= a.settle_ms and len(recent) == 3 and max(recent) <= 1.015 * min(recent):
                break
            if el >= 8.0 * a.settle_ms:
                break
        settle["ms"], settle["last_pass_ms"] = round(el, 1), [round(v, 4) for v in recent]
    # how many passes over the ring make one step: the K timed steps must last long enough that a few milliseconds of one-off host
    # latency (first launch after an idle queue, the wake-up of the final synchronize) cannot move the number -- VERDICT r4 item 1
    cycles, probe_ms = a.cycles, None
    if cycles <= 0:
        p0, p1 = F.Event(dev), F.Event(dev)
        mark_begin(p0)
        ring_pass()
        ring_pass()
        mark_end(p1, 2)
        p1.synchronize()
        probe_ms = p0.elapsed_ms(p1) / 2.0
        cycles = max(1, int(np.ceil(a.min_timed_ms / (a.steps * max(probe_ms, 1e-3)))))
        if dist is not None:            # every rank must run the same number of launches
            import torch
            tc = torch.tensor([cycles], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cycles = int(tc[0])

    def step():
        for _ in range(cycles):
            ring_pass()

    # the host's own cost of one launch, on a queue that never fills: 64 calls behind a synchronize (outside the timed region)
    sync()
    pairs, n_host = list(zip(srcs, dsts)), 64
    th0 = time.perf_counter()
    for i in range(n_host):
        s, d = pairs[i % len(pairs)]
        L.dcp_unwarp_image_f32(s.ptr, d.ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf, a.order, 1, blend, F.MEM_DEVICE, dev, None)
    host_call_us = (time.perf_counter() - th0) * 1e6 / n_host
    sync()

    for _ in range(a.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    e0, e1 = F.Event(dev), F.Event(dev)
    t0 = time.perf_counter()
    mark_begin(e0)
    for _ in range(a.steps):
        step()
    mark_end(e1)
    t_enq = time.perf_counter()            # every launch of the K steps is in the queue (the call blocks while the queue is full)
    sync()
    t_sync = time.perf_counter()
    if dist is not None:
        dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    enqueue_ms, sync_ms = (t_enq - t0) * 1e3, (t_sync - t_enq) * 1e3
    dev_ms = e0.elapsed_ms(e1)            # HIP events on the launch stream(s): device time of the K steps, gaps between launches included
    headline_kernel = F.last_kernel()
    if dist is not None:
        import torch
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(tt[0]), float(tt[1])

    # ---- everything below is outside the timed region
    verified = None
    copy_gbps = None
    if rank == 0:
        try:                   # frame 0 of the timed launches against the oracle (kernel arithmetic order, the same blend)
            orc = oracle_module(a.cpu_threads)
            ob = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32lerp": orc.BLEND_F32LERP}[a.blend]
            want = orc.unwarp_image_backward(img0, cfg["xcenter"], cfg["ycenter"], cfg["list_fact"], order=a.order, poly=orc.POLY_KERNEL,
                                             blend=ob)
            verified = bool(np.array_equal(download(dsts[0].ptr, (H, W), dev), want))
        except Exception as e:      # noqa: BLE001
            verified = "error: %r" % (e,)
        # context for roofline.frac: what a plain device-to-device copy of the same frames reaches on this box
        try:
            nbytes = H * W * 4
            scratch = F.DeviceBuffer(nbytes, dev)
            c0, c1 = F.Event(dev), F.Event(dev)
            for rep in range(2):
                if rep == 1:
                    c0.record()
                for _ in range(4):
                    for s in srcs:
                        F.check(L.dcp_memcpy(scratch.ptr, s.ptr, nbytes, F.COPY_D2D, dev, None))
            c1.record()
            c1.synchronize()
            copy_gbps = round(2.0 * nbytes * 4 * len(srcs) / (c0.elapsed_ms(c1) * 1e-3) / 1e9, 1)
            scratch.free()
        except Exception:      # noqa: BLE001
            copy_gbps = None

    dist_launch = None
    if rank == 0:
        try:          # per-launch distribution of the headline call (outside the timed region): the mean hides a few slow launches
            def one(i):
                F.check(L.dcp_unwarp_image_f32(srcs[i % a.batch].ptr, dsts[i % a.batch].ptr, H, W, W, 1, cfg["xcenter"], cfg["ycenter"], fa, nf,
                                               a.order, 1, blend, F.MEM_DEVICE, dev, None))
            dist_launch = launch_distribution(one, max(96, 4 * a.batch), dev)
        except Exception as e:      # noqa: BLE001 -- context only
            dist_launch = {"error": repr(e)}

    box = clocks_under_load(ring_pass, sync) if (rank == 0 and n_gpus == 1 and not a.no_extras) else None

    batched = None
    if rank == 0 and n_gpus == 1 and a.order == 1 and not a.no_extras:
        # context, not the metric: the frames of the ring share one calibration and lie in one (batch, H, W) array, so the PUBLIC batch
        # entry point (dcp_unwarp_images_f32: one pointer, centre and coefficient vector per frame) recognises them as the projections
        # of a stack and runs stack_wg_kernel -- bit-identical output, the coordinates evaluated once per pixel position instead of
        # once per frame.  The headline frames were written by per-frame launches: keep one to compare with.
        try:
            import ctypes as C
            n = a.batch
            ref = download(dsts[n - 1].ptr, (H, W), dev)
            dsts[n - 1].upload(np.zeros((H, W), np.float32))          # so that the comparison below cannot pass on stale pixels
            sp = (C.c_void_p * n)(*[b.ptr for b in srcs])
            dp = (C.c_void_p * n)(*[b.ptr for b in dsts])
            xa, ya = (C.c_double * n)(*([cfg["xcenter"]] * n)), (C.c_double * n)(*([cfg["ycenter"]] * n))
            table = np.ascontiguousarray([cfg["list_fact"]] * n, dtype=np.float64)
            tp = table.ctypes.data_as(C.POINTER(C.c_double))

            def batch_step(_i):
                F.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xa, ya, tp, nf, a.order, 1, blend, F.MEM_DEVICE, dev, None))
            per_frame_us = timed_launches(batch_step, max(4, min(a.steps, 40) // 4), dev) / n
            kb = F.last_kernel()
            chk = download(dsts[n - 1].ptr, (H, W), dev)
            batched = {"what": "the %d frames of the ring in ONE dcp_unwarp_images_f32 call (one calibration, frames of one array: "
                               "routed to the stack kernel)" % n,
                       "Mpixels_per_s": round(H * W / per_frame_us, 1), "us_per_frame": round(per_frame_us, 3),
                       "frac_of_hbm_peak": round(configs.BYTES_PER_PIXEL * H * W / (per_frame_us * 1e-6) / 1e9 / configs.HBM_PEAK_GBPS, 4),
                       "kernel": kb, "identical_to_per_frame_launches": bool(np.array_equal(chk, ref))}
        except Exception as e:      # noqa: BLE001 -- context only, never fails the bench
            batched = {"error": repr(e)}

    batched_distinct = None
    if rank == 0 and n_gpus == 1 and not a.no_extras:
        # beside the per-launch headline, not instead of it: the frames of a step through ONE dcp_unwarp_images_f32 call, every
        # frame with its OWN calibration (centre shifted, coefficients rescaled per frame) -- remap_wg_batch_kernel, the frame in
        # blockIdx.z, so that the drain of one frame runs under the ramp of the next
        try:
            batched_distinct = batched_distinct_calibrations(a, dev, srcs, dsts, img0, cfg, blend)
        except Exception as e:      # noqa: BLE001 -- context only, never fails the bench
            batched_distinct = {"error": repr(e)}

    others = None
    if rank == 0 and n_gpus == 1 and not a.no_extras:
        try:
            others = other_configs(a, dev, srcs, dsts, img0)
        except Exception as e:      # noqa: BLE001 -- context only
            others = {"error": repr(e)}
    ring_src.free()                 # the ring is no longer needed: make room for the 8192^2 frames and the stack
    ring_dst.free()
    if others is not None and "error" not in others:
        for name, fn in (("cfg5_frame8192_radial9", lambda: config5(a, dev)), ("color_4096x3", lambda: color_frame(a, dev)), ("cfg4_one_sinogram", lambda: stack_one_gpu_cases(a, dev)),
                         ("cfg4_grid_search_121_centres", lambda: grid_search_centres(a, dev)),
                         ("cfg2_uint16_frames_batched", lambda: uint16_frames_batched(a, dev)),
                         ("cfg4_uint16_shard64", lambda: stack_typed_shard(a, dev, "uint16")),
                         ("cfg4_int32_shard64", lambda: stack_typed_shard(a, dev, "int32")), ("cfg4_float64_shard64", lambda: stack_typed_shard(a, dev, "float64"))):
            try:
                others[name] = fn()
            except Exception as e:      # noqa: BLE001
                others[name] = {"error": repr(e)}

    scaling = None
    if not a.no_extras:
        try:
            depth = a.depth
            if world == 1:            # the whole stack needs 2 x 53.7 GB; fall back to one 8-GPU shard if that does not fit
                fr = free_device_bytes(dev)
                if fr is not None and fr < 2.2 * depth * 2560 * 2560 * 4:
                    depth = 256
            scaling = stack_on_hip(a, world, rank, dev, dist, backend, depth, 2560, steps=3, warmup=1, blend=F.BLEND_F64LERP)
            if others is not None and "error" not in others and world == 1:
                D_, _, W_ = configs.cfg4(depth)["shape"]
                ms = scaling["compute_only"]["ms_per_step"]
                others["cfg4_stack_one_gpu"] = entry(ms * 1e3, D_ * 2560 * W_, 8, scaling["kernel"], scaling["verified_vs_oracle"],
                                                     shape=[D_, 2560, W_], note="every row of every projection in one launch")
        except Exception as e:      # noqa: BLE001
            scaling = {"error": repr(e)}
        all_ok = isinstance(scaling, dict) and "error" not in scaling
        if world > 1:          # what follows is collective: every rank must take the same branch
            import torch
            flag = torch.tensor([1 if all_ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            all_ok = bool(int(flag[0]))
        if world > 1 and all_ok:
            try:
                scaling["rccl"] = torch_rccl_report(dist, backend, world, rank, dev_index, torch_dbg_dir)
                if "error" in scaling["rccl"]:          # the entry fails, not the bench
                    scaling["compute_plus_allgather"] = {"error": scaling["rccl"]["error"]}
            except Exception as e:      # noqa: BLE001
                scaling["rccl"] = {"error": repr(e)}
        # the exchange without torch (native RCCL through the C ABI; peer copies): child processes of rank 0, the other ranks wait
        # (on a one-GPU test box -- DCP_BENCH_DEVICE set -- only when DCP_RCCL_PATH names the tests' stand-in for librccl: RCCL
        # itself refuses two ranks on one device)
        if world > 1 and (os.environ.get("DCP_BENCH_DEVICE") is None or os.environ.get("DCP_RCCL_PATH")) and all_ok:
            sync()
            # the other ranks wait for rank 0's children on the HOST (the job's key-value store): an RCCL barrier would park a spinning
            # kernel on every other GPU while the children are being timed on those very GPUs
            store = None
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                store = _get_default_store()
            except Exception:      # noqa: BLE001
                store = None
            if rank == 0:
                try:
                    scaling["without_torch"] = native_exchange_variants(a, world, a.depth)
                except Exception as e:      # noqa: BLE001
                    scaling["without_torch"] = {"error": repr(e)}
                if store is not None:
                    try:
                        store.set("dcp_native_variants_done", "1")
                    except Exception:      # noqa: BLE001
                        store = None
            elif store is not None:
                try:
                    import datetime
                    store.wait(["dcp_native_variants_done"], datetime.timedelta(seconds=1800))
                except Exception:      # noqa: BLE001
                    pass
            if dist is not None:
                dist.barrier()

    if rank == 0:
        launches = a.steps * cycles * a.batch
        pix_per_launch = H * W
        total_pix = launches * pix_per_launch * n_gpus
        value_wall = total_pix / wall / 1e6
        value = total_pix / (dev_ms * 1e-3) / 1e6
        launch_us = dev_ms * 1e3 / launches
        achieved = configs.BYTES_PER_PIXEL * pix_per_launch / (launch_us * 1e-6) / 1e9
        traffic, traffic_source, traffic_stale = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("kernel", "").split("<")[0] == headline_kernel.split("<")[0]:
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_stale = j.get("kernel_sources_sha256") != kernel_sources_digest()
                    traffic_source = "profiles/pmc_latest.json: rocprofv3 PMC passes of %s over this command (%s, commit %s), not measured in this run" % (
                        j.get("kernel"), j.get("collected", "date not recorded"), j.get("git_commit", "not recorded"))
                else:
                    traffic_source = "none: profiles/pmc_latest.json is of %s, this run launched %s" % (j.get("kernel"), headline_kernel)
            except Exception:      # noqa: BLE001
                traffic = None
        out = {
            "metric": "Mpixels/s backward unwarp (4096x4096, 5-term poly, bilinear)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall * 1e3 / a.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            # value: from the HIP events on the launch stream around the K steps (max over ranks; the gaps between launches are
            # inside); value_wall / ms_per_step: host clock between the two barrier + synchronize pairs (max over ranks)
            "value_basis": "device events around the K timed steps; value_wall is the same work over the host's wall clock",
            "value_wall": round(value_wall, 1), "ms_per_step_device": round(dev_ms / a.steps, 4),
            "timed_region_ms": round(wall * 1e3, 3), "launches_timed_per_gpu": launches,
            # rank 0's host side of the timed region: the launch loop (it blocks while the queue is full, so this tends to the
            # device time on long regions), the final synchronize, and one launch call on a queue that never fills
            "host_enqueue_us_per_launch": round(enqueue_ms * 1e3 / launches, 3), "sync_ms": round(sync_ms, 3),
            "host_call_us_per_launch_idle_queue": round(host_call_us, 3),
            "launch_us": round(launch_us, 3),
            # which dispatch mode produced `value`: the per-frame entry point, one calibration per call, in every mode
            "dispatch": {"mode": a.dispatch,
                         "what": {"two_streams": "the ring's frames alternate over two streams made with dcp_stream_create -- two independent "
                                                 "frames in flight, the drain of one under the ramp of the other; launch_us = timed region / launches",
                                  "ordered": "one stream: every launch starts when the previous one's last workgroup has retired",
                                  "any_order": "one stream, DCP_MEM_DEVICE_UNORDERED: dispatch packets without the barrier bit"}[a.dispatch],
                         "single_stream_launch_us": None if not isinstance(dist_launch, dict) else dist_launch.get("back_to_back_mean_us"),
                         "ab": "profiles/r06a_dispatch_modes.txt, profiles/r06b_dispatch_streams.txt (tools/time_dispatch.py: ordered / any-order / "
                               "2..8 streams / one batched launch on two boxes)"},
            "launch_us_median": None if not isinstance(dist_launch, dict) else dist_launch.get("median_us"),
            "launch_us_p90": None if not isinstance(dist_launch, dict) else dist_launch.get("p90_us"),
            "data": "synthetic (numpy default_rng uniform [0,1) float32 frames, device-resident)",
            "config": {"workload": cfg["name"], "frames_per_step_per_gpu": cycles * a.batch, "distinct_frames_in_ring": a.batch,
                       "ring_passes_per_step": cycles, "ring_pass_probe_ms": None if probe_ms is None else round(probe_ms, 4), "height": H, "width": W,
                       "nfact": nf, "order": a.order, "blend": a.blend, "coord_round_f32": True, "pixel_dtype": "f32",
                       "arithmetic": "coordinates and blend in float64 (as numpy / scipy compute them), pixels float32; the f64lerp "
                                     "blend is a factorisation within one float32 ulp of scipy's operation order (other_configs."
                                     "cfg2_scipy_exact_blend is the bit-equal mode)", "clock_settle_ms": a.settle_ms, "clock_settle": settle,
                       "parallelism": "independent frames per GPU (no collective)" if n_gpus > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": configs.HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / configs.HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         # true: the kernel sources were edited after the counters were collected (SHA-256 over discorpy_amd/csrc)
                         "traffic_stale": traffic_stale, "kernel": headline_kernel, "launch_us": round(launch_us, 3),
                         "dispatch": a.dispatch,
                         # the same launches on ONE stream, back to back (no events in between), measured after the timed region: what
                         # `rocprofv3 --kernel-trace --stats` of an --dispatch ordered run reports as the kernel's average duration;
                         # with two frames in flight rocprofv3's per-kernel duration is about twice launch_us (two kernels overlap)
                         "single_stream": None if not (isinstance(dist_launch, dict) and dist_launch.get("back_to_back_mean_us")) else {
                             "launch_us": dist_launch["back_to_back_mean_us"],
                             "frac": round(configs.BYTES_PER_PIXEL * pix_per_launch / (dist_launch["back_to_back_mean_us"] * 1e-6) / 1e9 / configs.HBM_PEAK_GBPS, 4)},
                         "algorithmic_bytes_per_launch": int(configs.BYTES_PER_PIXEL * pix_per_launch),
                         "d2d_copy_same_frames_GBps": copy_gbps, "per_launch_distribution": dist_launch},
            "verified_vs_oracle": verified,
        }
        # the other BASELINE configurations as top-level scalars too (whatever keeps only part of this line keeps more than one
        # number): cfg3 fused perspective o radial, cfg4 whole stack on one GPU, cfg5 8192^2 9-term, the colour frame
        if isinstance(others, dict) and "error" not in others:
            def pick(name, key, scale=1.0):
                v = others.get(name)
                return None if not isinstance(v, dict) or key not in v else round(v[key] * scale, 4)
            out["cfg3_fused_us"], out["cfg3_fused_frac"] = pick("cfg3_fused", "launch_us"), pick("cfg3_fused", "frac")
            out["cfg4_stack_ms"], out["cfg4_stack_frac"] = pick("cfg4_stack_one_gpu", "launch_us", 1e-3), pick("cfg4_stack_one_gpu", "frac")
            out["cfg5_us"], out["cfg5_frac"] = pick("cfg5_frame8192_radial9", "launch_us"), pick("cfg5_frame8192_radial9", "frac")
            out["color_4096x3_us"], out["color_4096x3_frac"] = pick("color_4096x3", "launch_us"), pick("color_4096x3", "frac")
            out["cubic_spline_us"] = pick("cfg2_order3_cubic_spline", "launch_us")
            # the bit-equal-to-scipy blend and order 0 of the headline workload (the headline itself is timed on the default f64lerp
            # blend, within one float32 ulp of scipy's operation order)
            out["cfg2_scipy_exact_us"], out["cfg2_scipy_exact_frac"] = pick("cfg2_scipy_exact_blend", "launch_us"), pick("cfg2_scipy_exact_blend", "frac")
            out["cfg2_order0_us"], out["cfg2_order0_frac"] = pick("cfg2_order0_nearest", "launch_us"), pick("cfg2_order0_nearest", "frac")
        if isinstance(batched, dict) and "error" not in batched:
            out["batched_same_calibration_us_per_frame"], out["batched_same_calibration_frac"] = batched["us_per_frame"], batched["frac_of_hbm_peak"]
        if box is not None:
            out["box"] = box
        if others is not None:
            out["other_configs"] = others
        if scaling is not None:
            out["stack_scaling"] = scaling
        if batched is not None:
            out["batched_same_calibration"] = batched
        if batched_distinct is not None:
            out["batched_distinct_calibrations"] = batched_distinct
        if e2e is not None:
            out["end_to_end_numpy"] = e2e
        if n_gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, img0, blend, a.cpu_threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
