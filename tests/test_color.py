"""Interleaved colour images on remap_wg_color_kernel (util.unwarp_color_image_backward, reference
discorpy/util/utility.py:278-342: map_coordinates channel by channel at ONE set of coordinates, :320-341).
Every comparison is bit for bit: against the oracle per channel, against golden G10 (the reference's own output), and against
three single-plane calls of unwarp_image_backward (remap_wg_kernel) under the same blend."""
import ctypes as C

import numpy as np
import pytest

from conftest import HOST, golden, noise, typed_image

pytestmark = pytest.mark.gpu

FACT5 = [1.00227490554, -9.3601153805625e-06, 8.78436609375e-09, -4.79328802218628e-12, 7.714082828693389e-16]


def planes_from_oracle(orc, rgb, xc, yc, fact, order, blend):
    ob = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}[blend]
    if rgb.dtype == np.float32:
        return np.stack([orc.unwarp_image_backward(np.ascontiguousarray(rgb[:, :, c]), xc, yc, fact, order=order, poly=orc.POLY_KERNEL, blend=ob)
                         for c in range(rgb.shape[2])], axis=2)
    yd, xd = orc.radial_coords(rgb.shape[0], rgb.shape[1], xc, yc, fact, poly=orc.POLY_KERNEL)
    return np.stack([orc.map_coordinates(np.ascontiguousarray(rgb[:, :, c]), yd, xd, order) for c in range(rgb.shape[2])], axis=2)


@pytest.mark.parametrize("shape, centre, fact, staged", [
    ((1000, 1536, 3), (700.3, 480.9), FACT5, True),
    ((517, 1031, 4), (500.0, 250.0), [1.0, -2e-5, 3e-8], True),     # ragged tiles on both axes, four channels
    ((40, 56, 3), (27.4, 19.1), [1.0, 3e-5, 3e-7], True),            # golden G10's shape: one partial tile
    ((40, 56, 3), (27.4, 19.1), [1.0, 4e-3, 5e-5], False),           # ... and a model too curved for the tile certificate
    ((300, 700, 3), (-50.0, 900.0), [0.98, 1e-5, 1e-8, 1e-12, 1e-15, 1e-18, 1e-21], True),   # seven terms: the NF = 10 instantiation, centre outside
])
def test_float32_colour_equals_the_oracle_and_three_single_plane_calls(hip, orc, shape, centre, fact, staged):
    from discorpy_amd.post import postprocessing as pp
    from discorpy_amd.util import utility as util
    rgb = noise(11, shape) * 255.0
    xc, yc = centre
    for order, blend in ((1, None), (1, "scipy"), (0, None)):
        got = util.unwarp_color_image_backward(rgb, xc, yc, fact, order=order, blend=blend)
        assert hip.last_kernel().startswith("remap_wg_color_kernel" if staged else "typed_channels_kernel"), hip.last_kernel()
        assert got.dtype == np.float32 and got.shape == rgb.shape
        want = planes_from_oracle(orc, rgb, xc, yc, fact, order, blend or HOST)
        assert np.array_equal(got, want), (order, blend, int((got != want).sum()))
        for c in range(shape[2]):           # what three K1 calls give
            assert np.array_equal(got[:, :, c], pp.unwarp_image_backward(np.ascontiguousarray(rgb[:, :, c]), xc, yc, fact, order=order, blend=blend))


def test_golden_g10_through_the_colour_entry(hip):
    """The reference's own outputs (tools/gen_golden.py imports discorpy.util.utility), on whichever kernel the calibration's certificate selects."""
    from discorpy_amd.util import utility as util
    g = golden("g10_color40x56x3")
    rgb = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    fa, nf = hip.fact_array(a[2])
    level = hip.lib().dcp_debug_tile_certificate(0, 40, 56, a[0], a[1], fa, nf, None)
    out = util.unwarp_color_image_backward(rgb, *a, blend="scipy")
    assert hip.last_kernel().startswith("remap_wg_color_kernel" if level >= 2 else "typed_channels_kernel"), (level, hip.last_kernel())
    assert np.array_equal(out, g["nopad"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb, *a, pad=(3, 5, 2, 7), pad_mode="edge", blend="scipy"), g["pad_3_5_2_7_edge"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb, *a, order=0, pad=4, pad_mode="reflect"), g["pad_4_reflect_order0"])


@pytest.mark.parametrize("dt, channels, width", [("uint8", 3, 1532), ("uint8", 4, 1001), ("uint16", 3, 1030), ("uint16", 4, 777)])
def test_integer_colour_blends_and_stores_as_scipy_does(hip, orc, dt, channels, width):
    from discorpy_amd.util import utility as util
    rgb = typed_image(dt, (600, width, channels), 5)
    xc, yc, fact = 610.2, 333.3, [1.0, -3e-5, 4e-8]
    for order in (1, 0):
        got = util.unwarp_color_image_backward(rgb, xc, yc, fact, order=order)
        assert hip.last_kernel().startswith("remap_wg_color_kernel"), hip.last_kernel()
        assert got.dtype == rgb.dtype
        assert np.array_equal(got, planes_from_oracle(orc, rgb, xc, yc, fact, order, "scipy")), (dt, channels, order)


def test_what_the_staged_kernel_declines_gives_the_same_values(hip, orc):
    """Rows that are not dword-aligned (uint8 x 3 of odd width), 2 / 5 channels, a pixel stride above the channel count, and an
    uncertified (folding) model go to the one-thread-per-pixel kernel: same arithmetic, same bits."""
    from discorpy_amd.util import utility as util
    xc, yc, fact = 410.2, 233.3, [1.0, -3e-5, 4e-8]
    odd = typed_image("uint8", (300, 801, 3), 6)
    got = util.unwarp_color_image_backward(odd, xc, yc, fact)
    assert hip.last_kernel().startswith("typed_channels_kernel")
    assert np.array_equal(got, planes_from_oracle(orc, odd, xc, yc, fact, 1, "scipy"))
    for ch in (2, 5):
        img = noise(7, (300, 640, ch))
        for blend in (None, "scipy"):
            got = util.unwarp_color_image_backward(img, xc, yc, fact, blend=blend)
            assert hip.last_kernel().startswith("typed_channels_kernel")
            assert np.array_equal(got, planes_from_oracle(orc, img, xc, yc, fact, 1, blend or HOST)), (ch, blend)
    rgba = noise(8, (300, 640, 4))
    view = rgba[:, :, :3]                                    # pixel stride 4, three channels
    got = util.unwarp_color_image_backward(view, xc, yc, fact)
    assert np.array_equal(got, planes_from_oracle(orc, np.ascontiguousarray(view), xc, yc, fact, 1, HOST))
    fold = [1.0, -4e-3, 6e-6]                                # folds inside the frame: no certificate
    img = noise(9, (400, 640, 3))
    got = util.unwarp_color_image_backward(img, 320.0, 200.0, fold)
    assert hip.last_kernel().startswith("typed_channels_kernel")
    assert np.array_equal(got, planes_from_oracle(orc, img, 320.0, 200.0, fold, 1, HOST))


def test_device_resident_4096_rgb_and_a_band_of_rows(hip, orc):
    """BASELINE config 2's geometry with three channels, device-resident through the C ABI; the corner tiles of this model
    magnify by more than the slab allows and take the kernel's direct gather."""
    L = hip.lib()
    H, W, NC = 4096, 4096, 3
    rgb = noise(21, (H, W, NC))
    cases = [(1883.8169650464, 1478.6964217312, FACT5),
             (2048.0, 2048.0, [1.0, 0.0, 5.96e-9])]           # x-magnification 1.10 at the corners: their boxes exceed 144 pixels
    dsrc = hip.DeviceBuffer(rgb.nbytes).upload(rgb)
    ddst = hip.DeviceBuffer(rgb.nbytes)
    for xc, yc, fact in cases:
        fa, nf = hip.fact_array(fact)
        for blend, ob in ((hip.BLEND_F64LERP, "f64lerp"), (hip.BLEND_SCIPY, "scipy")):
            hip.check(L.dcp_unwarp_color_image(dsrc.ptr, ddst.ptr, 0, H, W, NC, W * NC, NC, xc, yc, fa, nf, 1, blend, hip.MEM_DEVICE, -1, None))
            hip.check(L.dcp_stream_synchronize(-1, None))
            assert hip.last_kernel().startswith("remap_wg_color_kernel<NF=5,%s,float32 x 3>" % ob), hip.last_kernel()
            got = ddst.download((H, W, NC), np.float32)
            want = planes_from_oracle(orc, rgb, xc, yc, fact, 1, ob)
            assert np.array_equal(got, want), (fact, ob, int((got != want).sum()))


@pytest.mark.parametrize("dt", ["float64", "int32", "uint32"])
def test_single_plane_frames_of_wide_element_types_on_the_staged_kernel(hip, orc, dt):
    """float64 / int32 / uint32 frames (they ran on the one-thread-per-pixel kernel): the interleaved-pixel
    kernel with ONE channel -- scipy's exact blend and store, bit-equal to the oracle; device-resident and host (banded) calls."""
    from discorpy_amd.post import postprocessing as pp
    for (h, w), (xc, yc), fact in (((700, 1100), (500.2, 333.3), [1.0, -3e-5, 4e-8]), ((2300, 2048), (1000.5, 1200.0), FACT5), ((33, 150), (70.0, 12.0), [1.01, 1e-5])):
        im = typed_image(dt, (h, w), 17)
        yd, xd = orc.radial_coords(h, w, xc, yc, fact, poly=orc.POLY_KERNEL)
        for order in (1, 0):
            got = pp.unwarp_image_backward(im, xc, yc, fact, order=order)
            assert hip.last_kernel().startswith("remap_wg_color_kernel") and (dt + " x 1") in hip.last_kernel(), hip.last_kernel()
            assert got.dtype == im.dtype and np.array_equal(got, orc.map_coordinates(im, yd, xd, order)), (dt, h, w, order)
    import torch
    t = torch.from_numpy(typed_image(dt, (600, 900), 18).astype(np.int64 if dt == "uint32" else dt)) if dt == "uint32" else torch.from_numpy(typed_image(dt, (600, 900), 18))
    if dt != "uint32":                 # (torch has no uint32 arithmetic type worth the detour)
        dev = pp.unwarp_image_backward(t.cuda(), 400.0, 300.0, [1.0, 2e-5])
        torch.cuda.synchronize()
        assert hip.last_kernel().startswith("remap_wg_color_kernel")
        yd, xd = orc.radial_coords(600, 900, 400.0, 300.0, [1.0, 2e-5], poly=orc.POLY_KERNEL)
        assert np.array_equal(dev.cpu().numpy(), orc.map_coordinates(t.numpy(), yd, xd, 1))


def test_sheared_map_on_64x32_workgroup_tiles_is_an_equal_and_slower_alternative(hip, orc):
    """a second workgroup-tile shape (64 x 32 under an 80 x 56 box) for maps whose 128 x 32 tiles are sheared
    out of the slab (BASELINE config 5's fisheye model).  Built and measured (tools/time_cfg5.py: 128-131 us against 113-116 for the
    per-wave-box kernel on the 8192^2 frame), so it stays an option (tall_tiles = 1); what it computes is bit-equal."""
    from discorpy_amd import configs
    from discorpy_amd.post import postprocessing as pp
    c5 = configs.cfg5()
    s = 8192 / 2048.0
    H = W = 2048
    fact = [a * s ** i for i, a in enumerate(c5["list_fact"])]          # the same model on a quarter-size frame
    xc, yc = c5["xcenter"] / s, c5["ycenter"] / s
    img = noise(41, (H, W))
    want = orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)
    host_blend = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}[HOST]
    try:
        hip.set_option("x_tall_tiles", 1)
        got = pp.unwarp_image_backward(img[:1024], xc, yc, fact)          # (a host frame below the banded path's threshold: one launch)
        assert "64x32 tiles" in hip.last_kernel(), hip.last_kernel()
        w2 = orc.unwarp_image_backward(np.ascontiguousarray(img[:1024]), xc, yc, fact, poly=orc.POLY_KERNEL, blend=host_blend)
        assert np.array_equal(got, w2)
        hip.set_option("x_tall_tiles", 0)
        assert np.array_equal(pp.unwarp_image_backward(img[:1024], xc, yc, fact), w2) and hip.last_kernel().startswith("remap_lds_kernel")
    finally:
        hip.set_option("x_tall_tiles", 0)
    assert want.shape == (H, W)
