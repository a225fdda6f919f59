"""Many frames in one launch: dcp_unwarp_images_f32 / post.unwarp_images_backward (remap_wg_batch_kernel, blockIdx.z = frame)
-- every frame with its OWN calibration must equal the oracle, and the single-frame entry point, bit for bit.
Reference behaviour restated: one call of unwarp_image_backward per image (postprocessing.py:111-148), looped over channels in
examples/readthedocs_demo/demo_06.py:111-113."""
import ctypes as C

import numpy as np
import pytest

from conftest import HOST, noise, oblend

from discorpy_amd import _ffi as F
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp
from discorpy_amd.util import utility as ut

ORC_BLEND = {"scipy": "BLEND_SCIPY", "f64lerp": "BLEND_F64LERP", "f32": "BLEND_F32LERP"}


def _calibrations(n, H, W, seed):
    """n distinct mild calibrations (all certified at level 2 on frames of this size), with coefficient vectors of 3..5 terms."""
    rng = np.random.default_rng(seed)
    c2 = configs.cfg2()
    s = min(4096.0 / max(H, W), 1.5)
    out = []
    for i in range(n):
        nf = 3 + i % 3
        fact = [c2["list_fact"][k] * (s ** k) * (1.0 + 0.1 * rng.standard_normal()) for k in range(nf)]
        fact[0] = 1.0 + 0.01 * rng.standard_normal()
        out.append((W * (0.3 + 0.4 * rng.random()), H * (0.3 + 0.4 * rng.random()), fact))
    return out


def _device_batch(hip, frames, cals, order, blend_code, dev=-1):
    """The C ABI on device pointers, exactly as bench.py calls it; returns the downloaded outputs and the kernel that ran."""
    L = hip.lib()
    n = len(frames)
    H, W = frames[0].shape
    src = [hip.DeviceBuffer(f.nbytes, dev).upload(f) for f in frames]
    dst = [hip.DeviceBuffer(f.nbytes, dev) for f in frames]
    nf = max(len(c[2]) for c in cals)
    table = np.zeros((n, nf))
    for i, c in enumerate(cals):
        table[i, :len(c[2])] = c[2]
    sp = (C.c_void_p * n)(*[b.ptr for b in src])
    dp = (C.c_void_p * n)(*[b.ptr for b in dst])
    xa = (C.c_double * n)(*[c[0] for c in cals])
    ya = (C.c_double * n)(*[c[1] for c in cals])
    hip.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xa, ya, table.ctypes.data_as(C.POINTER(C.c_double)), nf, order, 1, blend_code,
                                      hip.MEM_DEVICE, dev, None))
    kernel = hip.last_kernel()
    outs = [b.download((H, W), np.float32) for b in dst]
    for b in src + dst:
        b.free()
    return outs, kernel


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["f64lerp", "scipy", "f32"])
def test_batch_of_distinct_calibrations_equals_the_oracle_per_frame(hip, orc, blend):
    H, W, n = 700, 1000, 7
    frames = [noise(100 + i, (H, W)) for i in range(n)]
    cals = _calibrations(n, H, W, 5)
    outs, kernel = _device_batch(hip, frames, cals, 1, hip.BLEND_BY_NAME[blend])
    assert kernel.startswith("remap_wg_batch_kernel<Radial,NF=5"), kernel
    for f, (xc, yc, fact), got in zip(frames, cals, outs):
        want = orc.unwarp_image_backward(f, xc, yc, fact, poly=orc.POLY_KERNEL, blend=getattr(orc, ORC_BLEND[blend]))
        assert np.array_equal(got, want)
        assert np.array_equal(got, pp.unwarp_image_backward(f, xc, yc, fact, blend=blend))     # == one call per image
    # (seven frames: the tile hulls came from box_table_kernel's table; the same batch with every wave evaluating its corners)
    old = hip.get_option("x_box_table")
    hip.set_option("x_box_table", 0)
    try:
        outs0, _ = _device_batch(hip, frames, cals, 1, hip.BLEND_BY_NAME[blend])
    finally:
        hip.set_option("x_box_table", old)
    assert all(np.array_equal(a, b) for a, b in zip(outs, outs0))


@pytest.mark.gpu
def test_batch_order0_and_long_coefficient_vectors(hip, orc):
    H, W, n = 300, 520, 4
    frames = [noise(200 + i, (H, W)) for i in range(n)]
    cals = _calibrations(n, H, W, 6)
    outs, kernel = _device_batch(hip, frames, cals, 0, hip.BLEND_SCIPY)
    assert kernel.startswith("remap_wg_batch_kernel<Radial,NF=5,nearest"), kernel
    for f, (xc, yc, fact), got in zip(frames, cals, outs):
        assert np.array_equal(got, orc.unwarp_image_backward(f, xc, yc, fact, order=0, poly=orc.POLY_KERNEL))
    # 6..10 coefficients: the NF = 10 instantiation (shorter vectors padded with zeros -- bit-identical)
    c5 = configs.cfg5()
    long_fact = [a * 2.0 ** k for k, a in enumerate(c5["list_fact"])]
    cals = [(W * 0.5, H * 0.5, long_fact), (W * 0.45, H * 0.52, long_fact[:7]), (W * 0.4, H * 0.6, long_fact[:2])]
    outs, kernel = _device_batch(hip, frames[:3], cals, 1, hip.BLEND_F64LERP)
    assert kernel.startswith("remap_wg_batch_kernel<Radial,NF=10"), kernel
    for f, (xc, yc, fact), got in zip(frames, cals, outs):
        assert np.array_equal(got, orc.unwarp_image_backward(f, xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))


@pytest.mark.gpu
def test_batch_longer_than_one_launch_and_uncertified_frames_fall_back(hip, orc):
    # 60 frames > the 55 one launch carries: two launches, every frame still its own calibration
    H, W, n = 96, 200, 60
    frames = [noise(300 + i, (H, W)) for i in range(n)]
    cals = _calibrations(n, H, W, 7)
    outs, kernel = _device_batch(hip, frames, cals, 1, hip.BLEND_F64LERP)
    assert kernel.startswith("remap_wg_batch_kernel"), kernel
    for i in (0, 1, 54, 55, 56, 59):
        xc, yc, fact = cals[i]
        assert np.array_equal(outs[i], orc.unwarp_image_backward(frames[i], xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))
    # one strongly curved (uncertified) calibration in the batch: the whole call goes frame by frame, results unchanged
    cals2 = list(cals[:4])
    cals2[2] = (W * 0.5, H * 0.5, [1.0, 4e-3, 3e-5])
    outs, kernel = _device_batch(hip, frames[:4], cals2, 1, hip.BLEND_F64LERP)
    assert not kernel.startswith("remap_wg_batch_kernel"), kernel
    for f, (xc, yc, fact), got in zip(frames, cals2, outs):
        assert np.array_equal(got, orc.unwarp_image_backward(f, xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))


@pytest.mark.gpu
def test_full_size_batch_three_calibrations_4096(hip, orc):
    """BASELINE config 2 frames, three calibrations, through the call bench.py times (`batched_distinct_calibrations`)."""
    c2 = configs.cfg2()
    H, W = c2["shape"]
    frames = [noise(c2["seed"] + i, (H, W)) for i in range(3)]
    cals = [(c2["xcenter"], c2["ycenter"], c2["list_fact"]),
            (c2["xcenter"] + 37.25, c2["ycenter"] - 11.5, [v * (1.0 + 0.05 * k) for k, v in enumerate(c2["list_fact"])]),
            (c2["xcenter"] - 150.0, c2["ycenter"] + 80.0, [c2["list_fact"][0] * 0.99, c2["list_fact"][1] * 0.5, c2["list_fact"][2] * 0.7])]
    outs, kernel = _device_batch(hip, frames, cals, 1, hip.BLEND_F64LERP)
    assert kernel == "remap_wg_batch_kernel<Radial,NF=5,f64lerp>", kernel
    for f, (xc, yc, fact), got in zip(frames, cals, outs):
        assert np.array_equal(got, orc.unwarp_image_backward(f, xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))


@pytest.mark.gpu
def test_python_front_end_numpy_torch_and_colour_planes(hip, orc):
    H, W, n = 260, 390, 3
    frames = [noise(400 + i, (H, W)) for i in range(n)]
    cals = _calibrations(n, H, W, 8)
    xcs, ycs, facts = [c[0] for c in cals], [c[1] for c in cals], [c[2] for c in cals]
    want = [orc.unwarp_image_backward(f, *c, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP) for f, c in zip(frames, cals)]
    want_host = [orc.unwarp_image_backward(f, *c, poly=orc.POLY_KERNEL, blend=oblend(orc, HOST)) for f, c in zip(frames, cals)]
    # NumPy frames (host path: frame by frame inside the C call), per-frame calibrations of different lengths
    got = pp.unwarp_images_backward(frames, xcs, ycs, facts)
    assert isinstance(got, list) and all(np.array_equal(g, w) for g, w in zip(got, want_host))
    # one 3-D array, one shared calibration
    got = pp.unwarp_images_backward(np.stack(frames), xcs[0], ycs[0], facts[0])
    assert got.shape == (n, H, W)
    for i in range(n):
        assert np.array_equal(got[i], orc.unwarp_image_backward(frames[i], *cals[0], poly=orc.POLY_KERNEL, blend=oblend(orc, HOST)))
    # other element types and spline orders go image by image, same results as the single calls
    u16 = [(f * 60000).astype(np.uint16) for f in frames]
    got = pp.unwarp_images_backward(u16, xcs, ycs, facts)
    assert all(np.array_equal(g, pp.unwarp_image_backward(u, *c)) for g, u, c in zip(got, u16, cals))
    got = pp.unwarp_images_backward(frames, xcs, ycs, facts, order=3, mode="mirror")
    assert all(np.array_equal(g, pp.unwarp_image_backward(f, *c, order=3, mode="mirror")) for g, f, c in zip(got, frames, cals))
    with pytest.raises(ValueError):
        pp.unwarp_images_backward(frames, xcs[:2], ycs, facts)
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        t = torch.from_numpy(np.stack(frames)).cuda()
        got = pp.unwarp_images_backward(t, xcs, ycs, facts)
        torch.cuda.synchronize()
        assert F.last_kernel().startswith("remap_wg_batch_kernel") and tuple(got.shape) == (n, H, W)
        assert all(np.array_equal(got[i].cpu().numpy(), want[i]) for i in range(n))
        got = pp.unwarp_images_backward([t[i] for i in range(n)], xcs, ycs, facts, blend="scipy")
        torch.cuda.synchronize()
        for i in range(n):
            assert np.array_equal(got[i].cpu().numpy(), orc.unwarp_image_backward(frames[i], *cals[i], poly=orc.POLY_KERNEL, blend=orc.BLEND_SCIPY))
        # the colour image's plane-by-plane path (blend="f32" is not offered by the interleaved kernel): one launch for the 3 planes
        rgb = torch.from_numpy(np.stack(frames, axis=2)).cuda()
        got = ut.unwarp_color_image_backward(rgb, xcs[0], ycs[0], facts[0], blend="f32")
        torch.cuda.synchronize()
        assert F.last_kernel().startswith("remap_wg_batch_kernel<Radial,NF=5,f32lerp"), F.last_kernel()
        for i in range(n):
            assert np.array_equal(got[:, :, i].cpu().numpy(),
                                  orc.unwarp_image_backward(frames[i], *cals[0], poly=orc.POLY_KERNEL, blend=orc.BLEND_F32LERP))


def test_front_end_validation_needs_no_gpu():
    a = np.zeros((4, 5), np.float32)
    with pytest.raises(ValueError):
        pp.unwarp_images_backward(a, 1.0, 1.0, [1.0])                       # a single 2-D image is not a batch
    with pytest.raises(ValueError):
        pp.unwarp_images_backward([a, a], [1.0, 2.0, 3.0], 1.0, [1.0])      # three centres for two images
    with pytest.raises(ValueError):
        pp.unwarp_images_backward([a, a], 1.0, 1.0, [[1.0], [1.0], [1.0]])  # three coefficient vectors for two images
    with pytest.raises(RuntimeError):
        pp.unwarp_images_backward([a], 1.0, 1.0, [1.0], mode="bogus")       # scipy's message for an unknown mode
    assert pp.unwarp_images_backward([], 1.0, 1.0, [1.0]) == []
    L = F.lib()
    one = (C.c_double * 1)(1.0)
    ptr = (C.c_void_p * 1)(a.ctypes.data)
    assert L.dcp_unwarp_images_f32(ptr, ptr, -1, 4, 5, 5, 1, one, one, one, 1, 1, 1, F.BLEND_F64LERP, F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_images_f32(None, ptr, 1, 4, 5, 5, 1, one, one, one, 1, 1, 1, F.BLEND_F64LERP, F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_images_f32(ptr, ptr, 1, 4, 5, 5, 1, one, one, one, 40, 1, 1, F.BLEND_F64LERP, F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_images_f32(ptr, ptr, 0, 4, 5, 5, 1, one, one, one, 1, 1, 1, F.BLEND_F64LERP, F.MEM_HOST, -1, None) == F.OK


@pytest.mark.gpu
def test_a_stacked_device_array_that_is_not_a_tensor_comes_back_as_one_3d_array(hip, orc):
    """unwarp_images_backward promised one 3-D array for a 3-D input and returned a list for DeviceArray / CuPy-style
    inputs."""
    from discorpy_amd.post import postprocessing as pp
    n, H, W = 5, 300, 520
    stack = noise(91, (n, H, W))
    dev = hip.DeviceArray((n, H, W), np.float32).copy_from_host(stack)
    xcs = [250.0 + i for i in range(n)]
    ycs = [150.0 - i for i in range(n)]
    fact = [1.0, -2e-5, 3e-8]
    res = pp.unwarp_images_backward(dev, xcs, ycs, fact)
    assert isinstance(res, hip.DeviceArray) and res.shape == (n, H, W)
    assert hip.last_kernel().startswith("remap_wg_batch_kernel"), hip.last_kernel()
    got = res.copy_to_host()
    for i in range(n):
        assert np.array_equal(got[i], orc.unwarp_image_backward(stack[i], xcs[i], ycs[i], fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))
        assert np.array_equal(res[i].copy_to_host(), got[i])


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["f64lerp", "scipy"])
def test_frames_of_one_calibration_in_one_array_take_the_stack_kernel(hip, orc, blend):
    """the reference's callers loop ONE calibration over the channels / frames of an array
    (demo_06.py:111-113, demo_07.py:25,60).  dcp_unwarp_images_f32 sees equal calibrations + a constant pitch + dense results and
    runs the frames as the projections of a stack (stack_wg_kernel: a tile's coordinates once for all frames) -- bit-identical
    to one call per frame.  A gap between the frames (pitch > frame) is fine; scattered results are not (the batch kernel)."""
    L = hip.lib()
    H, W, n = 2048, 2304, 5              # (enough tiles for the stack kernel: its launcher declines launches of < 1024 workgroups)
    c2 = configs.cfg2()
    s = 4096.0 / W
    xc, yc = c2["xcenter"] / s, c2["ycenter"] / s
    fact = [v * s ** k for k, v in enumerate(c2["list_fact"])]
    frames = np.stack([noise(300 + i, (H, W)) for i in range(n)])
    gap = 4096                                   # elements between the end of one frame and the start of the next
    pitch = H * W + gap
    host = np.zeros(n * pitch, np.float32)
    for i in range(n):
        host[i * pitch:i * pitch + H * W] = frames[i].ravel()
    src = hip.DeviceBuffer(host.nbytes).upload(host)
    dst = hip.DeviceBuffer(frames.nbytes)
    fa = (C.c_double * (n * len(fact)))(*(fact * n))
    xa, ya = (C.c_double * n)(*([xc] * n)), (C.c_double * n)(*([yc] * n))
    sp = (C.c_void_p * n)(*[src.ptr + 4 * i * pitch for i in range(n)])
    dp = (C.c_void_p * n)(*[dst.ptr + 4 * i * H * W for i in range(n)])
    code = hip.BLEND_BY_NAME[blend]
    hip.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xa, ya, fa, len(fact), 1, 1, code, hip.MEM_DEVICE, -1, None))
    assert hip.last_kernel().startswith("stack_wg_kernel<NF=5," + blend), hip.last_kernel()
    got = dst.download((n, H, W), np.float32)
    for i in range(n):
        want = orc.unwarp_image_backward(frames[i], xc, yc, fact, poly=orc.POLY_KERNEL, blend=getattr(orc, ORC_BLEND[blend]))
        assert np.array_equal(got[i], want)
        assert np.array_equal(got[i], pp.unwarp_image_backward(frames[i], xc, yc, fact, blend=blend))
    # results NOT dense (frame 1 and 2 swapped): the frame-per-blockIdx.z kernel, the same pixels
    dp2 = (C.c_void_p * n)(*[dst.ptr + 4 * i * H * W for i in (0, 2, 1, 3, 4)])
    hip.check(L.dcp_unwarp_images_f32(sp, dp2, n, H, W, W, 1, xa, ya, fa, len(fact), 1, 1, code, hip.MEM_DEVICE, -1, None))
    assert hip.last_kernel().startswith("remap_wg_batch_kernel"), hip.last_kernel()
    got2 = dst.download((n, H, W), np.float32)
    assert all(np.array_equal(got2[j], got[i]) for i, j in enumerate((0, 2, 1, 3, 4)))
    # one centre differs in the last bit: not one calibration any more
    xb = (C.c_double * n)(*([xc] * (n - 1) + [np.nextafter(xc, 1e9)]))
    hip.check(L.dcp_unwarp_images_f32(sp, dp, n, H, W, W, 1, xb, ya, fa, len(fact), 1, 1, code, hip.MEM_DEVICE, -1, None))
    assert hip.last_kernel().startswith("remap_wg_batch_kernel"), hip.last_kernel()
    src.free()
    dst.free()


@pytest.mark.gpu
def test_unwarp_images_backward_on_a_3d_array_of_one_calibration(hip, orc):
    """The Python entry on a (n, H, W) device array with ONE centre / coefficient vector: the stack kernel, one 3-D result."""
    torch = pytest.importorskip("torch")
    H, W, n = 2048, 2200, 3
    c2 = configs.cfg2()
    s = 4096.0 / W
    xc, yc = c2["xcenter"] / s, c2["ycenter"] / s
    fact = [v * s ** k for k, v in enumerate(c2["list_fact"])]
    frames = np.stack([noise(400 + i, (H, W)) for i in range(n)])
    t = torch.from_numpy(frames).cuda()
    out = pp.unwarp_images_backward(t, xc, yc, fact)
    assert hip.last_kernel().startswith("stack_wg_kernel<NF=5,f64lerp"), hip.last_kernel()
    assert tuple(out.shape) == (n, H, W)
    got = out.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], orc.unwarp_image_backward(frames[i], xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["uint16", "int32", "float64"])
def test_a_3d_device_array_of_another_element_type_takes_the_stack_kernel(hip, orc, dt):
    """Frames of a detector (uint16) or of any other element type in one (n, H, W) device array under one calibration: the stack
    kernel of that type (scipy's blend and store, as frame by frame), every frame equal to its single call and to the oracle."""
    torch = pytest.importorskip("torch")
    if not hasattr(torch, dt):
        pytest.skip("torch has no %s" % dt)
    H, W, n = 2048, 2200, 3
    c2 = configs.cfg2()
    s = 4096.0 / W
    xc, yc = c2["xcenter"] / s, c2["ycenter"] / s
    fact = [v * s ** k for k, v in enumerate(c2["list_fact"])]
    frames = np.stack([(noise(500 + i, (H, W)) * (60000.0 if dt != "float64" else 1.0)).astype(dt) for i in range(n)])
    t = torch.from_numpy(frames.view(np.int16) if dt == "uint16" and not hasattr(torch, "uint16") else frames).cuda()
    out = pp.unwarp_images_backward(t, xc, yc, fact)
    tag = {"uint16": "16-bit", "int32": "32-bit", "float64": "float64"}[dt]
    assert hip.last_kernel().startswith("stack_wg_kernel<NF=5,scipy," + tag), hip.last_kernel()
    got = out.cpu().numpy()
    assert got.shape == (n, H, W) and got.dtype == frames.dtype
    for i in range(n):
        assert np.array_equal(got[i], orc.unwarp_image_backward(frames[i], xc, yc, fact, poly=orc.POLY_KERNEL))
    one = pp.unwarp_image_backward(t[1], xc, yc, fact)
    assert np.array_equal(one.cpu().numpy(), got[1])


@pytest.mark.gpu
def test_a_3d_device_array_under_a_folding_model_gives_what_the_single_calls_give(hip, orc):
    """the stack shortcut used the chunk function's semantics -- crop to the row band [yd_min, yd_max) spanned
    by the first and last rows and reflect inside it (postprocessing.py:289-312) -- for FRAMES, whose reference is
    unwarp_image_backward: clip to the whole image (:144-145).  Under a model that folds rows the two differ, or the band is empty
    (the call failed).  Frames now travel with coord_round_f32 = 2 (whole-frame clip, no band) -- the float32 route inside the C ABI
    and the typed route of the front end alike -- and device arrays the stack entry point cannot address in place (column-strided
    views) go frame by frame instead of raising."""
    torch = pytest.importorskip("torch")
    H, W, n = 300, 420, 3
    frames = np.stack([(noise(900 + i, (H, W)) * 60000.0).astype(np.uint16) for i in range(n)])
    ff = np.stack([noise(910 + i, (H, W)) for i in range(n)])
    L = hip.lib()
    for xc, yc, fact in ((210.0, 150.0, [1.0, -9e-3, 1.5e-5]),       # folds: B changes sign inside the frame
                         (200.0, 140.0, [-1.0, 0.0]),                # point reflection: rows reversed, first row maps below the last
                         (500.0, 900.0, [0.0, 0.0, 1e-9])):          # everything onto the centre row, outside the frame: a band of (nearly) nothing
        t = torch.from_numpy(frames.view(np.int16) if not hasattr(torch, "uint16") else frames).cuda()
        out = pp.unwarp_images_backward(t, xc, yc, fact).cpu().numpy().view(np.uint16)
        for i in range(n):
            want = orc.unwarp_image_backward(frames[i], xc, yc, fact, poly=orc.POLY_KERNEL)
            if hasattr(torch, "uint16"):
                assert np.array_equal(out[i], want), (fact, i)
            assert np.array_equal(out[i].view(t.cpu().numpy().dtype), pp.unwarp_image_backward(t[i], xc, yc, fact).cpu().numpy()), (fact, i)
        # float32 frames through dcp_unwarp_images_f32 (frames_as_stack declines or runs them with the whole-frame clip; never an error)
        tf = torch.from_numpy(ff).cuda()
        gotf = pp.unwarp_images_backward(tf, xc, yc, fact).cpu().numpy()
        for i in range(n):
            assert np.array_equal(gotf[i], orc.unwarp_image_backward(ff[i], xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)), (fact, i)
        # the C entry point itself with coord_round_f32 = 2 against per-frame oracle results, and = 1 (chunk semantics) untouched
        src = hip.DeviceBuffer(ff.nbytes).upload(ff)
        dst = hip.DeviceBuffer(ff.nbytes)
        fa, nf = hip.fact_array(fact)
        hip.check(L.dcp_unwarp_stack_rows_f32(src.ptr, dst.ptr, n, H, W, H * W, W, xc, yc, fa, nf, 0.0, H, 2, hip.BLEND_F64LERP, hip.MEM_DEVICE, -1, None))
        assert np.array_equal(dst.download((n, H, W), np.float32), gotf), fact
        src.free()
        dst.free()
    # a column-strided 3-D device view (one channel of interleaved frames): frame by frame, not a ValueError
    inter = torch.from_numpy(np.stack([frames, frames[::-1]], axis=-1).astype(np.int16 if not hasattr(torch, "uint16") else np.uint16)).cuda()
    view = inter[:, :, :, 0]
    assert view.stride(2) == 2
    got = pp.unwarp_images_backward(view, 210.0, 150.0, [1.0, 1e-4])
    for i in range(n):
        assert torch.equal(got[i], pp.unwarp_image_backward(view[i].contiguous(), 210.0, 150.0, [1.0, 1e-4]))


@pytest.mark.gpu
def test_independent_frames_over_two_streams_and_with_unordered_packets(hip, orc):
    """Round 6: callers that hand over INDEPENDENT frames one call at a time may spread them over two streams
    (dcp_stream_create + dcp_stream_wait_event to fork from / join into one of them: bench.py's default dispatch) or mark the calls
    DCP_MEM_DEVICE_UNORDERED (dispatch packets without the barrier bit).  Either way every frame is what the ordered call gives --
    the oracle's pixels -- and events / synchronisations on the streams still cover every launch."""
    L = hip.lib()
    c2 = configs.cfg2()
    H, W, n = 1536, 2048, 6
    s = 4096.0 / W
    xc, yc = c2["xcenter"] / s, c2["ycenter"] / s
    fact = [v * s ** k for k, v in enumerate(c2["list_fact"])]
    coef = [0.98, -0.012, 20.5, 0.009, 1.01, -14.0, 4e-6, -3e-6]
    frames = [noise(700 + i, (H, W)) for i in range(n)]
    src = [hip.DeviceBuffer(f.nbytes).upload(f) for f in frames]
    dst = [hip.DeviceBuffer(f.nbytes) for f in frames]
    fa, nf = hip.fact_array(fact)
    ca, _ = hip.fact_array(coef)
    want = [orc.unwarp_image_backward(f, xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP) for f in frames]
    s0, s1 = hip.Stream(), hip.Stream()
    e0, e1, ej = hip.Event(), hip.Event(), hip.Event()
    for mem, streams in ((hip.MEM_DEVICE, (s0, s1)), (hip.MEM_DEVICE_UNORDERED, (s0, s0)), (hip.MEM_DEVICE_UNORDERED, (s0, s1))):
        for d in dst:
            d.upload(np.zeros((H, W), np.float32))
        e0.record(s0.ptr)
        s1.wait_event(e0)                               # fork: nothing on s1 starts before e0
        for rep in range(3):
            for i in range(n):
                hip.check(L.dcp_unwarp_image_f32(src[i].ptr, dst[i].ptr, H, W, W, 1, xc, yc, fa, nf, 1, 1, hip.BLEND_F64LERP, mem, -1, streams[i & 1].ptr))
        ej.record(s1.ptr)
        s0.wait_event(ej)                               # join: e1 on s0 lies behind every launch of both streams
        e1.record(s0.ptr)
        e1.synchronize()                                # ONE wait for everything
        assert e0.elapsed_ms(e1) > 0.0
        for i in range(n):
            assert np.array_equal(dst[i].download((H, W), np.float32), want[i]), (mem, i)
    # the perspective and fused entry points take the flag too
    hip.check(L.dcp_perspective_image_f32(src[0].ptr, dst[0].ptr, H, W, W, 1, ca, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE_UNORDERED, -1, s0.ptr))
    hip.check(L.dcp_unwarp_fused_f32(src[1].ptr, dst[1].ptr, H, W, W, 1, xc, yc, fa, nf, ca, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE_UNORDERED, -1, s0.ptr))
    s0.synchronize()
    assert np.array_equal(dst[0].download((H, W), np.float32), orc.correct_perspective_image(frames[0], coef, blend=orc.BLEND_F64LERP))
    assert np.array_equal(dst[1].download((H, W), np.float32), orc.unwarp_fused(frames[1], xc, yc, fact, coef, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))
    # ... and nothing else does: a stack call with it is refused, not silently ordered
    rc = L.dcp_unwarp_stack_rows_f32(src[0].ptr, dst[0].ptr, 1, H, W, H * W, W, xc, yc, fa, nf, 0.0, 8, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE_UNORDERED, -1, None)
    assert rc == hip.ERR_INVALID_ARG and "mem_kind" in hip.last_error()
    assert L.dcp_stream_wait_event(s0.ptr, None) == hip.ERR_INVALID_ARG
    for b in src + dst:
        b.free()
