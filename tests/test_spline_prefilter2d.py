"""spline_prefilter2d_kernel (round 5): both axes of the one-pole B-spline prefilter in one launch -- what
`order=2` / `order=3` of unwarp_image_backward / correct_perspective_image (discorpy/post/postprocessing.py:111,147,462,491;
order=3 in examples/readthedocs_demo/demo_07.py:60) run on float32 frames whose lines are long enough for the one-pass kernels.
Against the oracle (scipy's serial recursion with its exact boundary values) and against the two launches it replaces."""
import numpy as np
import pytest

from conftest import noise
from discorpy_amd import configs
from discorpy_amd.post import postprocessing as pp

pytestmark = pytest.mark.gpu

FUSED = "spline_prefilter2d_kernel + spline_wg_kernel<order=%d>"
TWO = "spline_col_lds_kernel + spline_row_lds_kernel + spline_wg_kernel<order=%d>"
PADDED = "spline_tile_filter_kernel + spline_row_lds_kernel + spline_wg_kernel<order=%d>"


def ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


@pytest.fixture
def options(hip):
    yield hip
    for key, val in (("x_spline_tiled", 1), ("x_spline_wg", 1), ("x_pf2d_chunk", 0), ("x_pf2d_xcd", 1), ("x_spline_xcd", 1), ("x_pf2d_two_pole", 1)):
        hip.set_option(key, val)


@pytest.mark.parametrize("shape", [(1100, 1347), (569, 571), (600, 2100), (2100, 700)])
def test_one_launch_prefilter_against_the_oracle_and_the_two_launches(options, orc, shape):
    """Partial stripes (184 / 200 columns) and partial steps (32 rows) on both axes, both one-pole orders, both boundary kinds of
    the filter, and the two modes that pad the image by 12 samples (edge values / zeros): the default takes the one-launch
    prefilter, is within the restart error of the path it replaces (<= 4 pixels one float32 ulp apart) and of the serial oracle
    (<= 8), whatever the rows per chunk and the tile order."""
    F = options
    c = configs.cfg2()
    img = noise(shape[0] + shape[1], shape)
    coef = [1.02, 0.015, -9.0, -0.012, 0.99, 6.0, 2.0e-6, -1.5e-6]
    for order, mode in [(3, "reflect"), (3, "mirror"), (2, "reflect"), (2, "grid-mirror"), (3, "nearest"), (2, "grid-constant"), (3, "grid-constant"),
                        (3, "constant"), (2, "wrap")]:
        a = (img, c["xcenter"] * shape[1] / 4096.0, 0.45 * shape[0], c["list_fact"])
        want = orc.unwarp_image_backward(*a, order=order, mode=mode, poly=orc.POLY_KERNEL)
        F.set_option("x_spline_tiled", 6)
        two = pp.unwarp_image_backward(*a, order=order, mode=mode)
        # ('nearest' and 'grid-constant' filter the image with 12 samples added per side: a float64 copy + one pass per axis there)
        assert F.last_kernel() == (PADDED if mode in ("nearest", "grid-constant") else TWO) % order, F.last_kernel()
        F.set_option("x_spline_tiled", 1)
        for chunk, xcd in ((0, 1), (64, 1), (96, 0)):
            F.set_option("x_pf2d_chunk", chunk)
            F.set_option("x_pf2d_xcd", xcd)
            got = pp.unwarp_image_backward(*a, order=order, mode=mode)
            assert F.last_kernel() == FUSED % order, F.last_kernel()
            d = ulps(got, want)
            assert d.max() <= 1 and np.count_nonzero(d) <= 8, (order, mode, chunk, int(d.max()), int(np.count_nonzero(d)))
            assert np.count_nonzero(got != two) <= 4, (order, mode, chunk)
        F.set_option("x_pf2d_chunk", 0)
        F.set_option("x_pf2d_xcd", 1)
        # the perspective map shares the prefilter
        got = pp.correct_perspective_image(img, coef, order=order, mode=mode)
        assert F.last_kernel() == FUSED % order, F.last_kernel()
        d = ulps(got, orc.correct_perspective_image(img, coef, order=order, mode=mode))
        assert d.max() <= 1 and np.count_nonzero(d) <= 8, (order, mode)


TWO_PASS = "spline_prefilter2d_kernel x 2 + spline_wg_kernel<order=%d>"
TILE = "spline_tile_filter_kernel x 2 + spline_wg_kernel<order=%d>"


@pytest.mark.parametrize("shape", [(1100, 1347), (900, 2100), (2100, 930)])
def test_two_pole_orders_take_one_pass_of_the_kernel_per_pole(options, orc, shape):
    """Round 6: orders 4 and 5 -- two poles -- as TWO passes of spline_prefilter2d_kernel, one pole each (image ->
    scratch plane with the first pole's long restart horizon -- its row-pass threads exchange causal values through the tile --,
    scratch plane -> coefficient plane with the second pole), instead of the tile filter's one launch per axis.  scipy applies P2 P1 along one axis and then along the other; the
    four operators commute, so the plane is the same up to the rounding of float64 sums: within one float32 ulp of the oracle on
    a handful of pixels, as at every restart of the one-pass kernels.  Padded modes, integer frames and an interleaved channel
    included; lines too short for z1^n to underflow (731 / 884 samples) keep the serial recursion."""
    F = options
    c = configs.cfg2()
    img = noise(shape[0] * 3 + shape[1], shape)
    coef = [1.02, 0.015, -9.0, -0.012, 0.99, 6.0, 2.0e-6, -1.5e-6]
    a = (img, c["xcenter"] * shape[1] / 4096.0, 0.45 * shape[0], c["list_fact"])
    for order, mode in [(4, "reflect"), (5, "reflect"), (5, "mirror"), (4, "grid-mirror"), (5, "nearest"), (4, "grid-constant"), (4, "constant"), (5, "wrap")]:
        want = orc.unwarp_image_backward(*a, order=order, mode=mode, poly=orc.POLY_KERNEL)
        F.set_option("x_pf2d_two_pole", 0)
        old = pp.unwarp_image_backward(*a, order=order, mode=mode)
        assert "prefilter2d" not in F.last_kernel(), F.last_kernel()
        F.set_option("x_pf2d_two_pole", 1)
        for chunk, xcd in ((0, 1), (96, 0)):
            F.set_option("x_pf2d_chunk", chunk)
            F.set_option("x_pf2d_xcd", xcd)
            got = pp.unwarp_image_backward(*a, order=order, mode=mode)
            assert F.last_kernel() == TWO_PASS % order, F.last_kernel()
            d = ulps(got, want)
            assert d.max() <= 1 and np.count_nonzero(d) <= 8, (order, mode, chunk, int(d.max()), int(np.count_nonzero(d)))
            assert np.count_nonzero(got != old) <= 8, (order, mode, chunk, int(np.count_nonzero(got != old)))
        F.set_option("x_pf2d_chunk", 0)
        F.set_option("x_pf2d_xcd", 1)
        got = pp.correct_perspective_image(img, coef, order=order, mode=mode)
        assert F.last_kernel() == TWO_PASS % order, F.last_kernel()
        d = ulps(got, orc.correct_perspective_image(img, coef, order=order, mode=mode))
        assert d.max() <= 1 and np.count_nonzero(d) <= 8, (order, mode)
    # integer frames and one channel of an interleaved image
    rng = np.random.default_rng(5)
    u16 = rng.integers(0, 65535, size=shape, endpoint=True, dtype=np.int64).astype(np.uint16)
    rgb = noise(77, shape + (3,))
    for order, src in ((4, u16), (5, u16), (5, rgb[:, :, 2])):
        got = pp.unwarp_image_backward(src, *a[1:], order=order)
        assert F.last_kernel() == TWO_PASS % order, F.last_kernel()
        want = orc.unwarp_image_backward(np.ascontiguousarray(src), *a[1:], order=order, poly=orc.POLY_KERNEL)
        if src.dtype == np.uint16:
            di = np.abs(got.astype(np.int64) - want.astype(np.int64))
            assert di.max() <= 1 and np.count_nonzero(di) <= 3, (order, int(di.max()), int(np.count_nonzero(di)))
        else:
            d = ulps(got, want)
            assert d.max() <= 1 and np.count_nonzero(d) <= 8, order
    # a 700-sample line: z1^n has not underflowed -- the chunked serial passes, as before
    short = noise(9, (700, 1200))
    pp.unwarp_image_backward(short, 600.0, 340.0, [1.0, -2e-5], order=5)
    assert "prefilter2d" not in F.last_kernel(), F.last_kernel()


def test_gather_tile_order_does_not_change_a_bit(options):
    """spline_wg_kernel deals its tiles to the XCDs in runs of neighbouring tile columns (remap_wg_kernel's order); in plain launch
    order (x_spline_xcd = 0) every pixel is the same -- widths that leave the last XCDs a tile column short included."""
    F = options
    for shape in ((700, 128 * 11 + 37), (640, 128 * 3), (600, 2100)):
        img = noise(shape[1], shape)
        a = (img, 0.52 * shape[1], 0.47 * shape[0], [1.0, -2e-5, 3e-8])
        coef = [1.01, 0.01, -5.0, -0.008, 0.99, 4.0, 1.5e-6, -1.0e-6]
        res = {}
        for xcd in (1, 0):
            F.set_option("x_spline_xcd", xcd)
            res[xcd] = (pp.unwarp_image_backward(*a, order=3), pp.correct_perspective_image(img, coef, order=2))
            assert "spline_wg_kernel" in F.last_kernel(), F.last_kernel()
        F.set_option("x_spline_xcd", 1)
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), shape


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16"])
def test_integer_frames_are_read_in_place(options, orc, dtype):
    """The integer element types cameras deliver take the one-launch prefilter too (their values are exact in float64: no float64 copy
    of the image first) -- plain, padded modes, an interleaved channel; results are scipy's rounded integers (a float64 sum on a half
    may round the other way: at most a handful of pixels one unit apart)."""
    F = options
    dt = np.dtype(dtype)
    info = np.iinfo(dt)
    rng = np.random.default_rng(41)
    shape = (700, 1100)
    img = rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=np.int64).astype(dt)
    rgb = rng.integers(info.min, info.max, size=shape + (3,), endpoint=True, dtype=np.int64).astype(dt)
    a = (0.47 * shape[1], 0.55 * shape[0], [1.0, -3e-5, 4e-8])
    for order, mode, src in [(3, "reflect", img), (2, "mirror", img), (3, "nearest", img), (2, "grid-constant", img), (3, "reflect", rgb[:, :, 1])]:
        got = pp.unwarp_image_backward(src, *a, order=order, mode=mode)
        assert F.last_kernel() == FUSED % order, F.last_kernel()
        want = orc.unwarp_image_backward(np.ascontiguousarray(src), *a, order=order, mode=mode, poly=orc.POLY_KERNEL)
        assert got.dtype == dt and want.dtype == dt
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and np.count_nonzero(d) <= 3, (dtype, order, mode, int(d.max()), int(np.count_nonzero(d)))
        F.set_option("x_spline_tiled", 6)
        old = pp.unwarp_image_backward(src, *a, order=order, mode=mode)
        assert "prefilter2d" not in F.last_kernel(), F.last_kernel()
        F.set_option("x_spline_tiled", 1)
        assert np.count_nonzero(got != old) <= 3, (dtype, order, mode)


def test_lines_too_short_for_the_one_pass_kernels_keep_the_serial_recursion(options, orc):
    """z^n has not underflowed on a 300-sample line (cubic: n > 565, quadratic: n > 423): the chunked passes, bit-equal to the oracle."""
    F = options
    img = noise(3, (300, 900))
    a = (img, 430.0, 160.0, [1.0, -2e-5, 3e-8])
    for order in (2, 3):
        got = pp.unwarp_image_backward(*a, order=order)
        assert "prefilter2d" not in F.last_kernel(), F.last_kernel()
        want = orc.unwarp_image_backward(*a, order=order, poly=orc.POLY_KERNEL)
        d = ulps(got, want)
        assert d.max() <= 1 and np.count_nonzero(d) <= 8


def test_channels_of_an_interleaved_image_and_row_bands_in_place(options, orc):
    """demo_06.py:111-113 / demo_07.py:25,60 loop over img[:, :, c] with order=3: the prefilter reads the channel at the pixel
    pitch; a row-band view of a taller frame (row stride > width) likewise."""
    F = options
    rgb = noise(11, (700, 900, 3))
    a = (433.3, 371.9, [1.0, -3e-5, 4e-8])
    for ch in range(3):
        view = rgb[:, :, ch]
        got = pp.unwarp_image_backward(view, *a, order=3)
        assert F.last_kernel() == FUSED % 3, F.last_kernel()
        want = orc.unwarp_image_backward(np.ascontiguousarray(view), *a, order=3, poly=orc.POLY_KERNEL)
        d = ulps(got, want)
        assert d.max() <= 1 and np.count_nonzero(d) <= 8, ch
    tall = noise(12, (900, 1024))
    band = tall[100:800, 50:950]                                 # row stride 1024, width 900
    got = pp.unwarp_image_backward(band, 440.0, 350.0, a[2], order=2, mode="mirror")
    assert F.last_kernel() == FUSED % 2, F.last_kernel()
    d = ulps(got, orc.unwarp_image_backward(np.ascontiguousarray(band), 440.0, 350.0, a[2], order=2, mode="mirror", poly=orc.POLY_KERNEL))
    assert d.max() <= 1 and np.count_nonzero(d) <= 8


def test_cfg2_frame_at_order_3(options, orc):
    """BASELINE config 2's frame (4096 x 4096, five coefficients) at order 3 on the device-resident call bench.py times."""
    F = options
    L = F.lib()
    c = configs.cfg2()
    H, W = c["shape"]
    img = np.random.default_rng(20260928).random((H, W), dtype=np.float32)
    src = F.DeviceBuffer(img.nbytes, -1).upload(img)
    dst = F.DeviceBuffer(img.nbytes, -1)
    fa, nf = F.fact_array(c["list_fact"])
    F.check(L.dcp_unwarp_image_spline_f32(src.ptr, dst.ptr, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, 3, 0, F.MEM_DEVICE, -1, None))
    assert F.last_kernel() == FUSED % 3, F.last_kernel()
    got = np.empty((H, W), np.float32)
    F.check(L.dcp_memcpy(got.ctypes.data, dst.ptr, got.nbytes, F.COPY_D2H, -1, None))
    want = orc.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], order=3, mode="reflect", poly=orc.POLY_KERNEL)
    d = ulps(got, want)
    assert d.max() <= 1 and np.count_nonzero(d) <= 32, (int(d.max()), int(np.count_nonzero(d)))
    src.free()
    dst.free()


def test_spline_frames_on_two_streams_keep_their_own_planes(options, orc):
    """Round 6: the library keeps TWO coefficient workspaces per device, so spline calls on two streams -- independent frames handed
    over alternately, bench.py's default dispatch -- do not wait for each other (the prefilter of one frame runs under the gather
    of the other); a third stream reuses the least recently used workspace behind an event of its last user.  Every frame is what
    the single-stream call gives."""
    F = options
    L = F.lib()
    c = configs.cfg2()
    H, W, n = 1200, 1500, 6
    fa, nf = F.fact_array(c["list_fact"])
    xc, yc = c["xcenter"] * W / 4096.0, 0.45 * H
    frames = [noise(800 + i, (H, W)) for i in range(n)]
    src = [F.DeviceBuffer(f.nbytes).upload(f) for f in frames]
    dst = [F.DeviceBuffer(f.nbytes) for f in frames]
    streams = [F.Stream(), F.Stream(), F.Stream()]
    for order in (3, 5):
        want = []
        for i in range(n):
            F.check(L.dcp_unwarp_image_spline_f32(src[i].ptr, dst[i].ptr, H, W, W, 1, xc, yc, fa, nf, order, 0, F.MEM_DEVICE, -1, streams[0].ptr))
            streams[0].synchronize()
            want.append(dst[i].download((H, W), np.float32))
            dst[i].upload(np.zeros((H, W), np.float32))
        assert max(ulps(want[0], orc.unwarp_image_backward(frames[0], xc, yc, c["list_fact"], order=order, poly=orc.POLY_KERNEL)).max(), 0) <= 1
        for ns in (2, 3):
            for rep in range(4):
                for i in range(n):
                    F.check(L.dcp_unwarp_image_spline_f32(src[i].ptr, dst[i].ptr, H, W, W, 1, xc, yc, fa, nf, order, 0, F.MEM_DEVICE, -1,
                                                          streams[(i + rep) % ns].ptr))
            for st in streams:
                st.synchronize()
            for i in range(n):
                assert np.array_equal(dst[i].download((H, W), np.float32), want[i]), (order, ns, i)
    F.release_scratch()                    # both workspaces go; the next call allocates again
    F.check(L.dcp_unwarp_image_spline_f32(src[0].ptr, dst[0].ptr, H, W, W, 1, xc, yc, fa, nf, 3, 0, F.MEM_DEVICE, -1, None))
    F.check(L.dcp_stream_synchronize(-1, None))
    for b in src + dst:
        b.free()
