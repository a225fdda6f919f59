"""Randomised differential test: HIP path (through the Python front end and the C ABI) against the CPU oracle in the
kernels' evaluation order -- bit for bit for orders 0/1, within one ulp on a few pixels for spline orders.

    python tools/fuzz_parity.py [cases] [seed] [--bounds]

--bounds: the run must be on the bounds-checking build (make -C discorpy_amd/csrc bounds; DCP_LIB_PATH=discorpy_amd/lib/
libdiscorpy_hip_bounds.so) and ends by asserting that no LDS tap of any staged kernel left its slab (dcp_debug_bounds).
A third of the cases are drawn so that they REACH the staged kernels (certified calibrations, float32 / 8- / 16- / 32-bit integer data, frames of
at least a few tiles): remap_wg_kernel, remap_wg_batch_kernel, stack_wg_kernel, remap_wg_color_kernel.

Shapes from 1 x 1 to ~1500 x 1500, centres inside and far outside the image, polynomial lengths 0..12 from mild to
folding maps (which exercise the LDS kernel's "box does not fit" / "vote failed" fallbacks), strong homographies,
strided sources, every blend mode, stacks, explicit coordinates, element types, spline orders and boundary modes.
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc                      # noqa: E402  (checker only)
from discorpy_amd import _ffi as F                    # noqa: E402
from discorpy_amd.post import postprocessing as pp    # noqa: E402

MODES = orc.MODES
BLENDS = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}
DTYPES = ["float32"] * 6 + ["float64", "uint8", "int8", "uint16", "int16", "uint32", "int32", "int64", "uint64", "bool"]
WIDE = ("int64", "uint64", "bool")


def rand_shape(rng):
    kind = rng.integers(0, 10)
    if os.environ.get("FUZZ_BIG") and kind >= 7:     # FUZZ_BIG=1: a third of the cases are 1500 .. 6000 pixels on a side
        return int(rng.integers(1500, 6000)), int(rng.integers(1500, 6000))
    if kind == 0:
        return int(rng.integers(1, 4)), int(rng.integers(1, 80))
    if kind == 1:
        return int(rng.integers(1, 80)), int(rng.integers(1, 4))
    if kind == 2:
        return int(rng.integers(600, 1500)), int(rng.integers(600, 1500))
    return int(rng.integers(2, 420)), int(rng.integers(2, 420))


def rand_image(rng, shape, dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.random(shape) * 400.0 - 100.0).astype(dt)
    if dt.kind == "b":
        return rng.random(shape) < 0.5
    info = np.iinfo(dt)
    if dt.itemsize == 8:
        im = rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=dt)
        flat = im.reshape(-1)
        flat[::5] = rng.integers(0, 1 << 30, size=flat[::5].shape).astype(dt)
        flat[2::9] = info.max
        return im
    return rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=np.int64).astype(dt)


def certified_fact(rng, h, w, xc, yc, lengths=None):
    """A calibration that holds the level-2 tile certificate (what the staged kernels need): mild coefficients, checked on the host."""
    R = float(np.hypot(h, w))
    for _ in range(20):
        n = int(rng.integers(1, 8)) if lengths is None else int(lengths[int(rng.integers(0, len(lengths)))])
        f = [1.0 + float(rng.uniform(-0.03, 0.03))] + [float(rng.uniform(-0.04, 0.04)) / R ** i for i in range(1, n)]
        fa, nf = F.fact_array(f)
        if F.lib().dcp_debug_tile_certificate(0, h, w, xc, yc, fa, nf, None) >= 2:
            return f
    return [1.0]


def rand_fact(rng, h, w):
    n = int(rng.integers(0, 13))
    R = float(np.hypot(h, w))
    style = rng.integers(0, 6)
    f = []
    for i in range(n):
        if i == 0:
            f.append({0: 1.0, 1: float(rng.uniform(0.7, 1.3)), 2: float(rng.uniform(0.9, 1.1)), 3: 0.0,
                      4: float(rng.uniform(-2, 3)), 5: 1.0}[int(style)])
        else:
            amp = {0: 0.05, 1: 0.2, 2: 0.02, 3: 0.5, 4: 1.5, 5: 0.0}[int(style)]
            f.append(float(rng.uniform(-amp, amp)) / R ** i)
    return f


def rand_coef(rng, h, w):
    style = rng.integers(0, 4)
    s = {0: 0.02, 1: 0.15, 2: 0.5, 3: 0.0}[int(style)]
    p = {0: 1e-5, 1: 2e-4, 2: 2e-3, 3: 0.0}[int(style)]
    return [1.0 + rng.uniform(-s, s), rng.uniform(-s, s), rng.uniform(-0.2, 0.2) * w,
            rng.uniform(-s, s), 1.0 + rng.uniform(-s, s), rng.uniform(-0.2, 0.2) * h,
            rng.uniform(-p, p), rng.uniform(-p, p)]


def same(a, b, order, what):
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if order <= 1:
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))))
            raise AssertionError("%s: %d pixels differ, first at %s: %r vs %r" % (what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))
        return
    if a.dtype.kind == "f":
        tol = 2e-6 if a.dtype == np.float32 else 1e-11
        scale = max(1.0, float(np.max(np.abs(b))))
        assert np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) <= tol * scale, what
    else:
        d = np.abs(a.astype(np.int64) - b.astype(np.int64))
        assert d.max() <= 1 and np.count_nonzero(d) <= max(3, a.size // 2000), what


def device_tensor(a):
    """A ROCm torch tensor of a NumPy array (None without torch: the case then runs on host arrays)."""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    except ImportError:
        pass
    return None


def to_host(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


def one_case(rng, k):
    kind = ["radial"] * 5 + ["persp", "fused", "stack", "coords", "spline", "color", "batch", "centres"]
    kind = kind[int(rng.integers(0, len(kind)))]
    if os.environ.get("FUZZ_ONLY"):             # FUZZ_ONLY=spline,color: only these case kinds (the other draws are still consumed)
        only = os.environ["FUZZ_ONLY"].split(",")
        kind = only[int(rng.integers(0, len(only)))]
    h, w = rand_shape(rng)
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    order = int(rng.integers(0, 2))
    blend = list(BLENDS)[int(rng.integers(0, 3))]
    xc, yc = float(rng.uniform(-0.3, 1.3) * w), float(rng.uniform(-0.3, 1.3) * h)
    if rng.integers(0, 5) == 0:
        xc, yc = float(round(xc)), float(round(yc))
    fact = rand_fact(rng, h, w)
    staged = rng.integers(0, 3) == 0 and not os.environ.get("FUZZ_ONLY")
    if os.environ.get("FUZZ_STAGED_KIND"):       # FUZZ_STAGED_KIND=stack: every case a staged case of that kind (a campaign aimed at one kernel family)
        staged = True
    if staged:
        # aimed at the staged kernels: a frame of several tiles, a certified calibration, an element type they take
        kind = ("radial", "radial", "batch", "stack", "color", "persp", "fused")[int(rng.integers(0, 7))]
        kind = os.environ.get("FUZZ_STAGED_KIND", kind)
        h, w = int(rng.integers(40, 900)), int(rng.integers(130, 1400))
        dt = ("float32", "float32", "float32", "uint8", "uint16", "int16", "int32", "uint32", "float64")[int(rng.integers(0, 9))]
        xc, yc = float(rng.uniform(0.1, 0.9) * w), float(rng.uniform(0.1, 0.9) * h)
        # (the fused map takes the workgroup-box kernel with 4 or 5 coefficients in the kernel arguments)
        fact = certified_fact(rng, h, w, xc, yc, (4, 5) if kind == "fused" else None)
        if blend == "f32" and kind == "color":
            blend = "f64lerp"
        for key in ("x_wg_box", "tile_cert"):
            F.set_option(key, 1)
        F.set_option("x_stack_wg", 2)
    okw = dict(poly=orc.POLY_KERNEL)
    tag = "case %d %s%s %dx%d %s order %d blend %s xc=%r yc=%r fact=%r" % (k, "staged " if staged else "", kind, h, w, dt, order, blend, xc, yc, fact)
    if os.environ.get("FUZZ_TRACE"):          # one line per case BEFORE it runs: what was running when a device fault ended the process
        print(tag, flush=True)
    f32 = dt == "float32"
    kw = dict(blend=blend) if f32 else {}
    if f32:
        okw["blend"] = BLENDS[blend]
    if kind == "batch":
        # several frames of one shape, every frame its own centre and coefficient vector, in one call (device-resident
        # float32 frames under certified calibrations share ONE launch; everything else goes frame by frame inside)
        n = int(rng.integers(1, 7))
        frames = [rand_image(rng, (h, w), dt) for _ in range(n)]
        mild = staged or rng.integers(0, 2) == 0
        cals = []
        for _ in range(n):
            fx, fy = float(rng.uniform(0.2, 0.8) * w), float(rng.uniform(0.2, 0.8) * h)
            ff = [1.0 + float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-2e-5, 2e-5)), float(rng.uniform(-2e-8, 2e-8))][:int(rng.integers(1, 4))] \
                if mild else rand_fact(rng, h, w)
            cals.append((fx, fy, ff))
        dev = [device_tensor(f_) for f_ in frames] if (f32 and (staged or rng.integers(0, 3) > 0)) else [None]
        src = dev if dev[0] is not None else frames
        got = pp.unwarp_images_backward(src, [c_[0] for c_ in cals], [c_[1] for c_ in cals], [c_[2] for c_ in cals], order=order, **kw)
        for f_, c_, g_ in zip(frames, cals, got):
            same(to_host(g_), orc.unwarp_image_backward(f_, c_[0], c_[1], c_[2], order=order, **okw), order, tag + " batch frame cal=%r" % (c_,))
    elif kind == "centres":
        d = int(rng.integers(1, 6))
        vol = rand_image(rng, (d, h, w), dt)
        nk = int(rng.integers(1, 9))
        xs = [xc + float(rng.uniform(-20, 20)) for _ in range(nk)]
        ys = [yc + float(rng.uniform(-20, 20)) for _ in range(nk)]
        src = device_tensor(vol) if (f32 and rng.integers(0, 2)) else None
        src = vol if src is None else src
        skw = dict(blend=blend) if (f32 and blend != "f32") else {}
        sokw = dict(okw, blend=BLENDS[blend]) if (f32 and blend != "f32") else {k_: v for k_, v in okw.items() if k_ != "blend"}
        if rng.integers(0, 2) or h < 2:
            idx = float(rng.uniform(-2, h + 1)) if rng.integers(0, 3) == 0 else int(rng.integers(0, h))
            got = to_host(pp.unwarp_slice_backward_centres(src, xs, ys, fact, idx, **skw))
            for q in range(nk):
                same(got[q], orc.unwarp_slice_backward(vol, xs[q], ys[q], fact, idx, **sokw), 1, tag + " centres slice %r centre %d" % (idx, q))
        else:
            r0 = int(rng.integers(0, h))
            r1 = int(rng.integers(r0, min(h, r0 + 40)))
            try:
                got = to_host(pp.unwarp_chunk_slices_backward_centres(src, xs, ys, fact, r0, r1, **skw))
            except ValueError as e:          # a model that folds the rows onto an empty band: refused like the single call
                if "empty band" not in str(e):
                    raise
                return kind
            for q in range(nk):
                same(got[q], orc.unwarp_chunk_slices_backward(vol, xs[q], ys[q], fact, r0, r1, **sokw), 1, tag + " centres chunk %d..%d centre %d" % (r0, r1, q))
    elif kind == "radial":
        img = rand_image(rng, (h, w), dt)
        if rng.integers(0, 4) == 0 and h > 2 and w > 2:       # strided / padded view
            big = rand_image(rng, (h + 3, 2 * w + 5), dt)
            img = big[1:h + 1, 2:2 * w + 2:2] if rng.integers(0, 2) else big[2:h + 2, 3:w + 3]
        want = orc.unwarp_image_backward(np.ascontiguousarray(img), xc, yc, fact, order=order, **okw)
        same(pp.unwarp_image_backward(img, xc, yc, fact, order=order, **kw), want, order, tag)
    elif kind == "persp":
        img = rand_image(rng, (h, w), dt)
        coef = rand_coef(rng, h, w)
        if staged:
            coef = [1.0 + rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(-0.05, 0.05) * w,
                    rng.uniform(-0.02, 0.02), 1.0 + rng.uniform(-0.02, 0.02), rng.uniform(-0.05, 0.05) * h, rng.uniform(-1e-5, 1e-5), rng.uniform(-1e-5, 1e-5)]
        ok2 = {k_: v for k_, v in okw.items() if k_ != "poly"}
        want = orc.correct_perspective_image(img, coef, order=order, **ok2)
        same(pp.correct_perspective_image(img, coef, order=order, **kw), want, order, tag + " coef=%r" % coef)
    elif kind == "fused":
        img = rand_image(rng, (h, w), dt)
        coef = rand_coef(rng, h, w)
        if staged:      # a mild (tame) homography whose image of the frame crosses the frame's edges: the inner clip cuts through tiles
            s_, p_ = (0.02, 1e-5) if rng.integers(0, 2) else (0.08, 6e-5)
            coef = [1.0 + rng.uniform(-s_, s_), rng.uniform(-s_, s_), rng.uniform(-0.1, 0.1) * w, rng.uniform(-s_, s_), 1.0 + rng.uniform(-s_, s_),
                    rng.uniform(-0.1, 0.1) * h, rng.uniform(-p_, p_), rng.uniform(-p_, p_)]
        got = pp.unwarp_perspective_fused(img, xc, yc, fact, coef, order=order, **kw)
        if f32:
            want = orc.unwarp_fused(img, xc, yc, fact, coef, order=order, **okw)
        else:
            py, px = pp.generate_fused_map((h, w), xc, yc, fact, coef)
            want = orc.map_coordinates(img, py, px, order)
        same(got, want, order, tag + " coef=%r" % coef)
    elif kind == "stack":
        h, w = max(h, 2), max(w, 2)
        if h * w > 250000:
            h, w = h // 3 + 2, w // 3 + 2
        d = int(rng.integers(1, 6))
        vol = rand_image(rng, (d, h, w), dt)
        r0 = int(rng.integers(0, h))
        r1 = int(rng.integers(r0, min(h, r0 + 40)))
        if staged:      # enough rows and projections for stack_wg_kernel, the calibration kept
            d = int(rng.integers(2, 9))
            vol = rand_image(rng, (d, h, w), dt)
            r0 = int(rng.integers(0, max(1, h - 33)))
            r1 = int(rng.integers(min(h - 1, r0 + 31), h))
        else:
            xc, yc = float(rng.uniform(0, w)), float(rng.uniform(0, h))
            fact = rand_fact(rng, h, w)[:8]
        tag += " rows %d..%d depth %d xc=%r yc=%r fact=%r" % (r0, r1, d, xc, yc, fact)
        try:
            got = pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, r0, r1, **kw)
        except ValueError as e:
            # a model that folds the chunk's rows onto an EMPTY band of source rows (yd_min >= yd_max): the reference would
            # hand scipy a zero-row array; library and oracle both refuse
            assert "empty band" in str(e), tag
            try:
                orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, r0, r1, **okw)
            except Exception:      # noqa: BLE001
                got = None
            else:
                raise AssertionError(tag + ": the library refused an empty band the oracle accepts")
        if got is not None:
            same(got, orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, r0, r1, **okw), 1, tag + " chunk")
        same(pp.unwarp_slice_backward(vol, xc, yc, fact, r0, **kw),
             orc.unwarp_slice_backward(vol, xc, yc, fact, r0, **okw), 1, tag + " slice")
    elif kind == "coords":
        img = rand_image(rng, (h, w), dt)
        n = int(rng.integers(0, 5000))
        cdt = np.float32 if rng.integers(0, 2) else np.float64
        ys = (rng.uniform(-0.1, 1.1, n) * (h - 1)).astype(cdt)
        xs = (rng.uniform(-0.1, 1.1, n) * (w - 1)).astype(cdt)
        if n > 10:
            ys[:5] = [0, h - 1, 0.5, h - 1.5 if h > 1 else 0, (h - 1) / 2.0]
            xs[:5] = [w - 1, 0, 0.5, w - 1.5 if w > 1 else 0, (w - 1) / 2.0]
        ok2 = {k_: v for k_, v in okw.items() if k_ != "poly"}
        import warnings
        with warnings.catch_warnings():      # (coordinates 10 % outside the image on purpose: treated as scipy treats them under the mode)
            warnings.simplefilter("ignore", RuntimeWarning)
            got = pp.remap_coordinates(img, ys, xs, order=order, **kw)
        same(got, orc.remap_coords(img, ys, xs, order=order, **ok2), order, tag)
    elif kind == "spline":
        if (os.environ.get("FUZZ_BIG") and rng.integers(0, 2)) or (not os.environ.get("FUZZ_BIG") and rng.integers(0, 8) == 0):
            # frames large enough for the one-pass tile prefilter (lines of >= ~900 samples) and many gather tiles
            h, w = int(rng.integers(900, 1700)), int(rng.integers(900, 1700))
        else:
            h, w = min(h, 300), min(w, 300)
        if dt in WIDE:                      # (orders >= 2 on 64-bit integers / bool: float64 noise of the recursive filter decides the
            dt = "int32"                     # stored integer near the overflow and near 1.0 -- tests/test_oracle_golden.py wide_close)
        img = rand_image(rng, (h, w), dt)
        so = int(rng.integers(2, 6))
        mode = MODES[int(rng.integers(0, 8))]
        xc, yc = float(rng.uniform(0, w)), float(rng.uniform(0, h))
        fact = [1.0] + [float(rng.uniform(-0.05, 0.05)) / float(np.hypot(h, w)) ** i for i in range(1, int(rng.integers(1, 5)))]
        tag = "case %d spline %dx%d %s order %d mode %s xc=%r yc=%r fact=%r" % (k, h, w, dt, so, mode, xc, yc, fact)
        if os.environ.get("FUZZ_TRACE"):
            print(tag, flush=True)
        same(pp.unwarp_image_backward(img, xc, yc, fact, order=so, mode=mode),
             orc.unwarp_image_backward(img, xc, yc, fact, order=so, mode=mode, poly=orc.POLY_KERNEL), so, tag)
    else:
        from discorpy_amd.util import utility as util
        if staged:
            c = int(rng.integers(3, 5))
            pad = 0
        else:
            h, w = min(max(h, 2), 300), min(max(w, 2), 300)
            c = int(rng.integers(1, 5))
            pad = int(rng.integers(0, 6))
            xc, yc = float(rng.uniform(0, w)), float(rng.uniform(0, h))
            fact = rand_fact(rng, h, w)[:6]
        rgb = rand_image(rng, (h, w, c), dt)
        got = util.unwarp_color_image_backward(rgb, xc, yc, fact, order=order, pad=pad, pad_mode="edge", **kw)
        padded = np.pad(rgb, [(pad, pad), (pad, pad), (0, 0)], mode="edge")
        for ch in range(c):
            want = orc.unwarp_image_backward(np.ascontiguousarray(padded[:, :, ch]), xc + pad, yc + pad, fact, order=order, **okw)
            same(np.ascontiguousarray(got[:, :, ch]), want, order, tag + " colour pad %d channel %d" % (pad, ch))
    return kind


def main():
    argv = [v for v in sys.argv[1:] if not v.startswith("--")]
    want_bounds = "--bounds" in sys.argv[1:]
    cases = int(argv[0]) if len(argv) > 0 else 400
    seed = int(argv[1]) if len(argv) > 1 else 20260928
    orc.build()
    orc.set_threads(min(32, orc.max_threads()))
    F.lib()
    F.require_device()
    rng = np.random.default_rng(seed)
    counts, kernels, t0 = {}, {}, time.time()
    F.debug_counters()
    checking = F.debug_bounds()[4] == 1
    if want_bounds and not checking:
        raise SystemExit("--bounds needs the checking build: make -C discorpy_amd/csrc bounds; DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_bounds.so")
    # every kernel a case launched, not only its last one: each C-ABI call of the front end passes through F.check
    launched, plain_check = set(), F.check

    def recording_check(rc):
        plain_check(rc)
        name = F.last_kernel()
        base = name.split("<")[0].split(" ")[0]
        if "<Fused" in name or "<Persp" in name:           # the map kind beside the kernel: which kernels the fused / perspective maps reach
            base += name.split(",")[0][len(base):] + ">"
        launched.add(base or "(spline / point kernels)")
    F.check = recording_check
    for k in range(cases):
        F.set_option("x_stack_lds", (1, 2, 0)[k % 3])       # stack cases: automatic choice / forced staged kernel / direct
        F.set_option("x_stack_wg", (2, 1, 2, 0)[k % 4])     # workgroup-box stack kernel: whenever eligible / automatic / off
        F.set_option("x_wg_box", 0 if k % 5 == 4 else 1)    # one box per workgroup, or per wave tile
        F.set_option("tile_cert", 0 if k % 7 == 6 else 1) # the host certificate, or the per-pixel vote
        launched.clear()
        kind = one_case(rng, k)
        counts[kind] = counts.get(kind, 0) + 1
        for name in launched:
            kernels[name] = kernels.get(name, 0) + 1
    nofit, vote = F.debug_counters()
    bounds = F.debug_bounds()
    for key in ("x_stack_lds", "x_stack_wg", "x_wg_box", "tile_cert"):
        F.set_option(key, 1)
    print("fuzz_parity: %d cases (seed %d) all equal in %.1f s: %s; cases in which each kernel ran: %s; LDS-kernel fallbacks exercised: "
          "%d tiles did not fit, %d tiles failed the vote" % (cases, seed, time.time() - t0, dict(sorted(counts.items())),
                                                              dict(sorted(kernels.items())), nofit, vote))
    if checking:
        print("bounds-checking build: %d LDS taps outside their slab%s" % (bounds[0], "" if not bounds[0] else
                                                                            " (first: byte %d of a %d-byte slab, site %d)" % bounds[1:4]))
        assert bounds[0] == 0, bounds


if __name__ == "__main__":
    main()
