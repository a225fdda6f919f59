"""CPU suite, part 1: oracle/unwarp_oracle.c is held bit-equal to the reference's outputs.

The vectors in tests/golden/ were produced by tools/gen_golden.py, which imports the reference
(discorpy 1.7.0 + scipy 1.15.3) in the build container.  ORC_POLY_NUMPY + ORC_BLEND_SCIPY is the
reference's own arithmetic order; ORC_POLY_KERNEL is the order the HIP kernels use and must land
on the same float32 coordinates.
"""
import numpy as np
import pytest

from conftest import G12B_DTYPES, G12_DTYPES, g12_inputs, golden, noise, ulp_diff, wide_image

POLYS = ["numpy", "kernel"]


def poly_of(orc, name):
    return orc.POLY_NUMPY if name == "numpy" else orc.POLY_KERNEL


@pytest.mark.parametrize("poly", POLYS)
def test_g1_reference_box_image(orc, poly):
    g = golden("g1_box64")
    for order in (0, 1):
        out = orc.unwarp_image_backward(g["mat"], g["xcenter"], g["ycenter"], g["list_fact"], order=order,
                                        poly=poly_of(orc, poly))
        assert np.array_equal(out, g["out_order%d" % order])


@pytest.mark.parametrize("poly", POLYS)
def test_g2_reference_slice_and_chunk(orc, poly):
    g = golden("g2_stripes10x64x64")
    vol = np.repeat(g["mat"][None], int(g["depth"]), axis=0)
    p = poly_of(orc, poly)
    s = orc.unwarp_slice_backward(vol, g["xcenter"], g["ycenter"], g["list_fact"], int(g["index"]), poly=p)
    assert s.dtype == np.float32 and np.array_equal(s, g["slice_out"])
    c = orc.unwarp_chunk_slices_backward(vol, g["xcenter"], g["ycenter"], g["list_fact"], int(g["start"]),
                                         int(g["stop"]), poly=p)
    assert c.shape == g["chunk_out"].shape and np.array_equal(c, g["chunk_out"])


def test_g3_reference_perspective(orc):
    g = golden("g3_perspective64")
    c1 = orc.correct_perspective_image(g["mat"], g["coef_backward"])
    assert np.array_equal(c1, g["cor_backward"])
    assert np.array_equal(orc.correct_perspective_image(c1, g["coef_forward"]), g["cor_forward_of_backward"])
    assert np.array_equal(orc.correct_perspective_image(g["mat"], g["coef_backward"], order=0),
                          g["cor_backward_order0"])


@pytest.mark.parametrize("poly", POLYS)
def test_g4_dot_pattern_05(orc, poly):
    g = golden("g4_dot_pattern_05")
    out = orc.unwarp_image_backward(g["crop_in"], g["crop_xcenter"], g["crop_ycenter"], g["list_fact"],
                                    poly=poly_of(orc, poly))
    assert np.array_equal(out, g["crop_out"])


@pytest.mark.parametrize("poly", POLYS)
def test_g4b_config1_full_frame(orc, poly):
    """BASELINE config 1 at full size (examples/example_02.py:81): the oracle on the decoded 800 x 1280 frame equals
    the reference's output bit for bit (SHA-256 of the float32 output, a lattice of its pixels, G4's eight rows)."""
    import hashlib
    g, g4 = golden("g4b_dot_pattern_05_full"), golden("g4_dot_pattern_05")
    img = g["frame_u8"].astype(np.float32)
    assert img.shape == (800, 1280) and np.array_equal(img[395:406], g4["full_in_rows_band"])
    out = orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"], poly=poly_of(orc, poly))
    assert np.array_equal(out[::32, ::32], g["out_lattice"])
    assert np.array_equal(out[g4["full_rows"]], g4["full_out_rows"])
    st = g4["full_stats"]
    assert (img.min(), img.max()) == (st[0], st[1]) and out[400, 640] == np.float32(213.037353515625)
    assert (out[400, 640], out[10, 10], out[799, 1279]) == (st[4], st[5], st[6])
    assert hashlib.sha256(out.tobytes()).digest() == g["out_sha256"].tobytes()
    out0 = orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"], order=0, poly=poly_of(orc, poly))
    assert hashlib.sha256(out0.tobytes()).digest() == g["out_order0_sha256"].tobytes()


@pytest.mark.parametrize("poly", POLYS)
@pytest.mark.parametrize("name", ["g5_cfg2_160", "g5_offcentre_150x200", "g5_cfg5_9term_144"])
def test_g5_configs_reduced(orc, name, poly):
    g = golden(name)
    img = noise(g["seed"], g["shape"])
    p = poly_of(orc, poly)
    yd, xd = orc.radial_coords(int(g["shape"][0]), int(g["shape"][1]), g["xcenter"], g["ycenter"], g["list_fact"],
                               poly=p)
    assert np.array_equal(yd.astype(np.float32), g["yd"]) and np.array_equal(xd.astype(np.float32), g["xd"])
    for order in (0, 1):
        out = orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"], order=order, poly=p)
        assert np.array_equal(out, g["out_order%d" % order])


@pytest.mark.parametrize("poly", POLYS)
def test_g6_stack_rows(orc, poly):
    g = golden("g6_stack3x800x1280")
    vol = noise(g["seed"], g["shape"])
    p = poly_of(orc, poly)
    a = (g["xcenter"], g["ycenter"], g["list_fact"])
    for r in g["rows"]:
        assert np.array_equal(orc.unwarp_slice_backward(vol, *a, int(r), poly=p), g["slice_%d" % r])
    assert np.array_equal(orc.unwarp_slice_backward(vol, *a, 400.5, poly=p), g["slice_frac_400p5"])
    for key, (s0, s1) in {"chunk_395_402": (395, 402), "chunk_0_2": (0, 2), "chunk_797_799": (797, 799)}.items():
        assert np.array_equal(orc.unwarp_chunk_slices_backward(vol, *a, s0, s1, poly=p), g[key])


def test_g15_folding_model_reflects_inside_the_reference_band(orc):
    """A non-monotone model: rows of the chunk leave the band [yd_min, yd_max) the reference crops from its first and last
    rows, and scipy reflects them inside the cropped array (postprocessing.py:289-312).  Reproduced since round 2: the
    oracle equals the reference on every pixel, and differs from sampling the whole projection exactly where the golden
    says the row coordinate leaves the band."""
    g = golden("g15_folding_chunk")
    vol = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), g["list_fact"], int(g["start"]), int(g["stop"]))
    out = orc.unwarp_chunk_slices_backward(vol, *a)
    outside = g["outside_band"]
    assert 0 < outside.sum() < outside.size
    assert np.array_equal(out, g["ref_out"])
    assert np.array_equal(out[:, ~outside], g["absolute_out"][:, ~outside])
    assert not np.array_equal(out[:, outside], g["absolute_out"][:, outside])
    # other element types (output dtype = input dtype), and a geometry whose rows leave the band far on the low side, where
    # the reference's float32 subtraction yd_mat - yd_min rounds
    assert np.array_equal(orc.unwarp_chunk_slices_backward(vol.astype(np.float64), *a), g["ref_out_f64"])
    assert np.array_equal(orc.unwarp_chunk_slices_backward((vol * 60000).astype(np.uint16), *a), g["ref_out_u16"])
    vol2 = noise(g["case2_seed"], g["case2_shape"])
    a2 = (float(g["case2_xcenter"]), float(g["case2_ycenter"]), g["case2_list_fact"], int(g["case2_rows"][0]), int(g["case2_rows"][1]))
    assert np.array_equal(orc.unwarp_chunk_slices_backward(vol2, *a2), g["case2_ref_out"])
    assert np.array_equal(orc.unwarp_chunk_slices_backward(vol2.astype(np.float64), *a2), g["case2_ref_out_f64"])


def test_g16_map_index_outside_the_image_follows_scipy_modes(orc):
    """Explicit coordinates outside the image at orders 0 / 1: the oracle restates scipy's boundary handling; pinned to the
    reference's own correct_perspective_image(map_index=..., mode=...) outputs (golden G16) and, more densely, to scipy
    itself (float32 and uint16 images, float32 and float64 coordinates)."""
    from scipy.ndimage import map_coordinates
    g = golden("g16_map_index_outside")
    mat = noise(g["seed"], g["shape"])
    modes = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")
    for mode in modes:
        for order in (0, 1):
            want = g["%s_o%d" % (mode.replace("-", "_"), order)].ravel()
            assert np.array_equal(orc.remap_coords(mat, g["ys"], g["xs"], order=order, mode=mode), want), (mode, order)
    rng = np.random.default_rng(7)
    img = noise(3, (70, 90))
    u16 = (img * 60000).astype(np.uint16)
    for shape_img in (img, img[:1, :], img[:, :1].copy()):
        hh, ww = shape_img.shape
        ys = (rng.random(3000) * hh * 9 - hh * 4).astype(np.float32)
        xs = (rng.random(3000) * ww * 9 - ww * 4).astype(np.float32)
        for mode in modes:
            for order in (0, 1):
                for yy, xx in ((ys, xs), (ys.astype(np.float64) * 1.0000001, xs.astype(np.float64))):
                    assert np.array_equal(orc.remap_coords(shape_img, yy, xx, order=order, mode=mode),
                                          map_coordinates(shape_img, (yy, xx), order=order, mode=mode)), (mode, order, shape_img.shape)
    for mode in modes:
        for order in (0, 1):
            assert np.array_equal(orc.map_coordinates(u16, ys[:2000] * 0 + (rng.random(2000) * 500 - 200).astype(np.float32),
                                                      xs[:2000] * 0 + (rng.random(2000) * 600 - 250).astype(np.float32), order, mode).shape, (2000,))
    yy = (rng.random(2000) * 500 - 200).astype(np.float32)
    xx = (rng.random(2000) * 600 - 250).astype(np.float32)
    for mode in modes:
        for order in (0, 1):
            assert np.array_equal(orc.map_coordinates(u16, yy, xx, order, mode), map_coordinates(u16, (yy, xx), order=order, mode=mode)), (mode, order)


def test_explicit_coordinates_outside_the_image_at_spline_orders_follow_scipy(orc):
    """Orders 2..5: scipy moves a coordinate outside the image into the extended image (or returns cval / evaluates in the
    padded plane for 'constant', 'nearest', 'grid-constant') before it evaluates the spline; the oracle restates that --
    compared with scipy itself, float32 and uint16 images, all eight modes."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(11)
    for shape in ((37, 41), (64, 20)):
        img = noise(shape[0], shape)
        u16 = (img * 60000).astype(np.uint16)
        h, w = shape
        ys = (rng.random(2500) * h * 7 - h * 3).astype(np.float32)
        xs = (rng.random(2500) * w * 7 - w * 3).astype(np.float32)
        ys[:80] = np.linspace(-14.0, h + 13.0, 80)
        xs[:80] = 17.3
        for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
            for order in (2, 3, 4, 5):
                ref = map_coordinates(img, (ys, xs), order=order, mode=mode)
                got = orc.remap_coords(img, ys, xs, order=order, mode=mode)
                assert np.count_nonzero(ref != got) <= 2 and np.max(np.abs(ref - got)) <= 1e-6, (shape, mode, order)
                ref16 = map_coordinates(u16, (ys, xs), order=order, mode=mode).astype(np.int64)
                got16 = orc.map_coordinates(u16, ys, xs, order, mode).astype(np.int64)
                assert np.count_nonzero(ref16 != got16) <= 2 and np.max(np.abs(ref16 - got16)) <= 1, (shape, mode, order, "uint16")


def test_spline_orders_on_tiny_images_equal_scipy(orc):
    """Images of 1 x n, n x 1, 2 x 2 ... 12 x 12 pixels, orders 2..5, eight modes, coordinates inside and outside: the
    oracle equals scipy to the last bit -- including scipy's reflect-prefilter initialisation, which reads a partial sum
    where the closed form wants sample 0 (visible only on lines of a few samples)."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(5)
    for shape in ((2, 2), (3, 5), (5, 64), (1, 30), (33, 1), (8, 9), (12, 12), (2, 40)):
        img = rng.random(shape, dtype=np.float32)
        h, w = shape
        ys = (rng.random(1200) * h * 7 - h * 3).astype(np.float32)
        xs = (rng.random(1200) * w * 7 - w * 3).astype(np.float32)
        ys[:400] = (rng.random(400) * (h - 1)).astype(np.float32)
        xs[:400] = (rng.random(400) * (w - 1)).astype(np.float32)
        for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
            for order in (2, 3, 4, 5):
                ref = map_coordinates(img, (ys, xs), order=order, mode=mode)
                got = orc.remap_coords(img, ys, xs, order=order, mode=mode)
                assert np.count_nonzero(ref != got) <= 1 and np.max(np.abs(ref - got)) <= 1e-6, (shape, mode, order)


def test_every_element_type_order_and_mode_against_scipy_inside_and_outside(orc):
    """orc.map_coordinates against scipy.ndimage.map_coordinates itself: eight element types, orders 0..5, eight modes,
    degenerate shapes, coordinates inside and up to three image sizes outside."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(9)
    for shape in ((1, 1), (1, 7), (2, 1), (3, 3), (4, 17), (16, 16)):
        h, w = shape
        ys = (rng.random(300) * h * 7 - h * 3).astype(np.float32)
        xs = (rng.random(300) * w * 7 - w * 3).astype(np.float32)
        ys[:100] = (rng.random(100) * (h - 1)).astype(np.float32)
        xs[:100] = (rng.random(100) * (w - 1)).astype(np.float32)
        for dt in (np.float32, np.float64, np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32):
            if np.dtype(dt).kind == "f":
                img = (rng.random(shape) * 2000 - 700).astype(dt)
            else:
                ii = np.iinfo(dt)
                img = rng.integers(ii.min, ii.max, size=shape, endpoint=True).astype(dt)
            for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
                for order in range(6):
                    ref = map_coordinates(img, (ys, xs), order=order, mode=mode)
                    got = orc.map_coordinates(img, ys, xs, order, mode)
                    if np.dtype(dt).kind == "f":
                        ok = np.allclose(ref, got, rtol=2e-6 if dt == np.float32 else 1e-11, atol=1e-4 if dt == np.float32 else 1e-9)
                    else:
                        ok = np.max(np.abs(ref.astype(np.int64) - got.astype(np.int64))) <= (0 if order <= 1 else 1)
                    assert ok, (shape, np.dtype(dt).name, mode, order)


def g17_cases():
    """The inputs of golden G17 (tools/gen_golden.py g17_case, same draws in the same order)."""
    modes = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")
    rng = np.random.default_rng(1717)
    for k in range(60):
        h, w = int(rng.integers(2, 70)), int(rng.integers(2, 70))
        img = rng.random((h, w), dtype=np.float32)
        xc, yc = float(rng.uniform(0, w)), float(rng.uniform(0, h))
        fact = [1.0, float(rng.uniform(-2e-3, 2e-3)), float(rng.uniform(-3e-5, 3e-5))]
        order = int(rng.integers(0, 6))
        mode = modes[int(rng.integers(0, 8))]
        coef = [1 + rng.uniform(-.05, .05), rng.uniform(-.05, .05), rng.uniform(-3, 3), rng.uniform(-.05, .05),
                1 + rng.uniform(-.05, .05), rng.uniform(-3, 3), rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4)]
        yield k, img, xc, yc, fact, order, mode, [float(c) for c in coef]


def test_g17_small_frames_every_order_and_mode_equal_the_reference(orc):
    """60 random frames of 2..69 pixels a side, random order 0..5 and boundary mode, radial and perspective: the oracle
    reproduces the reference's outputs bit for bit (the spline prefilter's boundary handling shows in every pixel here)."""
    g = golden("g17_small_frames_orders_modes")
    for k, img, xc, yc, fact, order, mode, coef in g17_cases():
        assert np.array_equal(orc.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode), g["radial_%02d" % k]), (k, order, mode)
        assert np.array_equal(orc.correct_perspective_image(img, coef, order=order, mode=mode), g["persp_%02d" % k]), (k, order, mode)


def g18_cases():
    """The inputs of golden G18 (tools/gen_golden.py g18_case, same draws in the same order)."""
    rng = np.random.default_rng(1818)
    for t in range(40):
        d, h, w = int(rng.integers(1, 3)), int(rng.integers(3, 70)), int(rng.integers(3, 70))
        dt = [np.float32, np.uint16, np.float64][t % 3]
        vol = rng.random((d, h, w))
        vol = (vol * 60000).astype(dt) if dt == np.uint16 else vol.astype(dt)
        xc, yc = float(rng.uniform(-0.2 * w, 1.2 * w)), float(rng.uniform(-0.2 * h, 1.2 * h))
        fact = [1.0 + float(rng.uniform(-.1, .1)), float(rng.uniform(-3e-3, 3e-3)), float(rng.uniform(-1e-4, 1e-4))]
        idx = [float(rng.uniform(-5, h + 5)), int(rng.integers(-3, h + 3)), float(rng.integers(0, h)) + 0.5][(t // 3) % 3]
        yield t, vol, xc, yc, fact, idx


def test_g18_slice_with_any_index_equals_the_reference(orc):
    """unwarp_slice_backward with a fractional, negative or too large `index` (the reference does not validate it), float32 /
    uint16 / float64 stacks: bit-equal to the reference."""
    g = golden("g18_slice_any_index")
    for t, vol, xc, yc, fact, idx in g18_cases():
        out = orc.unwarp_slice_backward(vol, xc, yc, fact, idx)
        assert out.dtype == np.float32 and np.array_equal(out, g["slice_%02d" % t]), (t, vol.dtype, idx)


def test_chunk_equals_image_rows_and_slice_differs(orc):
    """chunk rows == image rows (float32 coordinates); slice keeps float64 ones."""
    g = golden("g6_stack3x800x1280")
    vol = noise(g["seed"], g["shape"])
    a = (g["xcenter"], g["ycenter"], g["list_fact"])
    img_out = orc.unwarp_image_backward(vol[1], *a)
    chunk = orc.unwarp_chunk_slices_backward(vol, *a, 395, 402)
    assert np.array_equal(chunk[1], img_out[395:403])
    sl = orc.unwarp_slice_backward(vol, *a, 400)
    assert not np.array_equal(sl[1], img_out[400])
    assert np.max(np.abs(sl[1] - img_out[400])) < 1e-3


@pytest.mark.parametrize("poly", POLYS)
def test_g7_fused_is_one_resampling(orc, poly):
    g = golden("g7_fused144")
    img = noise(g["seed"], g["shape"])
    p = poly_of(orc, poly)
    fused = orc.unwarp_fused(img, g["xcenter"], g["ycenter"], g["list_fact"], g["list_coef"], poly=p)
    assert np.array_equal(fused, g["fused_out"])
    two = orc.correct_perspective_image(orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"],
                                                                  poly=p), g["list_coef"])
    assert np.array_equal(two, g["twopass_out"])
    assert np.array_equal(orc.correct_perspective_image(img, g["list_coef"]), g["persp_out"])
    assert np.max(np.abs(fused - two)) > 0.05          # a different operator, not a rounding variant


@pytest.mark.parametrize("poly", POLYS)
def test_g8_clipping_stress(orc, poly):
    g = golden("g8_clip120x180")
    assert float(g["clipped_fraction"]) > 0.4
    img = noise(g["seed"], g["shape"])
    for order in (0, 1):
        out = orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"], order=order,
                                        poly=poly_of(orc, poly))
        assert np.array_equal(out, g["out_order%d" % order])


def test_g9_explicit_coordinates(orc):
    g = golden("g9_points33x47")
    img = noise(g["seed"], g["shape"])
    for order in (0, 1):
        assert np.array_equal(orc.remap_coords(img, g["ys"], g["xs"], order=order), g["out_order%d" % order])
        assert np.array_equal(orc.remap_coords(img, g["ys64"], g["xs64"], order=order), g["out64_order%d" % order])


def test_cheaper_blends_stay_within_their_stated_bounds(orc):
    """f64lerp: <= 1 float32 ulp of scipy's result; f32lerp: <= 2 ulp of the largest tap."""
    img = (noise(5, (97, 131)) * 255).astype(np.float32)
    a = (61.3, 40.2, [1.02, -3e-4, 2e-6])
    ref = orc.unwarp_image_backward(img, *a, blend=orc.BLEND_SCIPY)
    f64 = orc.unwarp_image_backward(img, *a, blend=orc.BLEND_F64LERP)
    f32 = orc.unwarp_image_backward(img, *a, blend=orc.BLEND_F32LERP)
    assert ulp_diff(ref, f64).max() <= 1
    assert np.max(np.abs(f32.astype(np.float64) - ref)) <= 2 * np.spacing(np.float32(255.0))


def test_strided_sources(orc):
    rgb = noise(7, (40, 50, 3))
    ch = rgb[:, :, 1]
    a = (24.0, 19.5, [1.0, 1e-3])
    assert np.array_equal(orc.unwarp_image_backward(np.ascontiguousarray(ch), *a),
                          orc.unwarp_image_backward(np.ascontiguousarray(ch).copy(), *a))
    padded = np.zeros((40, 64), np.float32)
    padded[:, :50] = ch
    assert np.array_equal(orc.unwarp_image_backward(padded[:, :50], *a),
                          orc.unwarp_image_backward(np.ascontiguousarray(ch), *a))


def test_degenerate_shapes(orc):
    for shape in [(1, 1), (1, 9), (9, 1), (2, 2), (2, 7)]:
        img = noise(3, shape)
        out = orc.unwarp_image_backward(img, 0.4 * shape[1], 0.6 * shape[0], [1.0, 1e-2])
        assert out.shape == shape and np.all(np.isfinite(out))
    one = np.full((1, 1), 3.5, np.float32)
    assert orc.unwarp_image_backward(one, 0.0, 0.0, [1.0])[0, 0] == np.float32(3.5)
    assert orc.unwarp_image_backward(one, 0.0, 0.0, [])[0, 0] == np.float32(3.5)   # empty list_fact: B = 0


# ---------------------------------------------------------------- spline orders 2..5

MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")


def spline_close(got, ref):
    """The recursive prefilter is restated, not copied, so its float64 coefficients differ from
    scipy's by ~1e-15 and the float32 result may land one ulp away on a few pixels in a thousand."""
    d = ulp_diff(got, ref)
    return d.max() <= 1 and np.count_nonzero(d) <= max(2, got.size // 500)


@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_g11_spline_orders_every_mode(orc, order):
    g = golden("g11_spline45x60")
    img = noise(g["seed"], g["shape"])
    for mode in MODES:
        out = orc.unwarp_image_backward(img, g["xcenter"], g["ycenter"], g["list_fact"], order=order, mode=mode)
        assert np.array_equal(out, g["radial_o%d_%s" % (order, mode)]), (order, mode)      # bit for bit since round 2
    assert np.array_equal(orc.remap_coords(img, g["pts_y"], g["pts_x"], order=order), g["points_o%d_reflect" % order])


def test_g11_perspective_order3_every_mode(orc):
    g = golden("g11_spline45x60")
    img = noise(g["seed"], g["shape"])
    for mode in MODES:
        assert np.array_equal(orc.correct_perspective_image(img, g["list_coef"], order=3, mode=mode), g["persp_o3_%s" % mode]), mode


def typed_close(out, ref, order):
    """Orders 0/1 are bit-exact; a spline order may flip a rounding where the double result sits on .5
    (integers) or differ in the last place (float64), as spline_close allows for float32."""
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if order <= 1 or out.dtype != np.float64:
        return np.array_equal(out, ref)          # spline orders too since round 2 (float32 and the integer types)
    return np.allclose(out, ref, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("dt", G12_DTYPES)
def test_g12_element_types(orc, dt):
    g = golden("g12_dtypes40x52")
    im, vol = g12_inputs(g, dt)
    xc, yc, fact, coef = g["xcenter"], g["ycenter"], g["list_fact"], g["list_coef"]
    for order in (0, 1, 3):
        assert typed_close(orc.unwarp_image_backward(im, xc, yc, fact, order=order), g["radial_o%d_%s" % (order, dt)], order)
        assert typed_close(orc.map_coordinates(im, g["pts_y"], g["pts_x"], order), g["points_o%d_%s" % (order, dt)], order)
    assert typed_close(orc.unwarp_image_backward(im, xc, yc, fact, order=2, mode="nearest"), g["radial_o2_nearest_" + dt], 2)
    assert typed_close(orc.correct_perspective_image(im, coef), g["persp_o1_" + dt], 1)
    assert typed_close(orc.correct_perspective_image(im, coef, order=5, mode="grid-wrap"), g["persp_o5_wrap_" + dt], 5)
    sl = orc.unwarp_slice_backward(vol, xc, yc, fact, int(g["index"]))
    assert sl.dtype == np.float32 and np.array_equal(sl, g["slice_" + dt])
    ch = orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, int(g["start"]), int(g["stop"]))
    assert typed_close(ch, g["chunk_" + dt], 1)


def wide_close(out, ref, order, precast=None):
    """Golden G12b: orders 0 / 1 bit for bit.  Order 3 on 64-bit integers compares doubles of magnitude ~1e19 after a separable
    recursive filter: the restatement and scipy agree to a few float64 ulps of the LARGEST values in the filter's support (1e-13
    of the data's range), which moves the stored integer -- and flips a result that lands within that distance of a threshold of
    the store: 2^63 / 2^64 / 0 between the type's range and what the overflowing cast stores, 1.0 for the truncating bool store.
    `precast` = the float64 result before the store (the oracle on the image read as doubles): ONLY pixels whose value lies at a
    threshold may disagree; every other pixel must agree -- equal for bool, to 1e-13 of the range for the integers (a
    5 % allowance over the whole image would have let a wrong edge band through)."""
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if order <= 1:
        return np.array_equal(out, ref)
    assert precast is not None and precast.shape == out.shape and precast.dtype == np.float64
    scale = max(float(np.max(np.abs(precast))), 1.0)
    if out.dtype == np.bool_:
        near = np.abs(np.abs(precast) - 1.0) <= 1e-9
        return bool(np.all((out == ref) | near)) and np.count_nonzero(out != ref) <= 0.12 * out.size
    thresholds = [-2.0 ** 63, 2.0 ** 63] if out.dtype == np.int64 else [0.0, 2.0 ** 63, 2.0 ** 64]
    near = np.zeros(out.shape, bool)
    for t in thresholds:
        near |= np.abs(precast - t) <= 4e-12 * scale
    o, r = out.astype(np.float64), ref.astype(np.float64)
    agree = np.abs(o - r) <= 1e-13 * max(float(np.max(np.abs(r))), 1.0)
    return bool(np.all(agree | near)) and np.count_nonzero(~agree) <= 0.12 * out.size          # (a ninth of the inputs sit AT the extremes)


@pytest.mark.parametrize("dt", G12B_DTYPES)
def test_g12b_int64_uint64_bool(orc, dt):
    """The element types scipy takes beyond the 32-bit ones: the oracle's restatement of the double read and of
    what the reference's undefined 64-bit casts store on x86-64, against the reference itself."""
    g = golden("g12b_wide_types40x52")
    im = wide_image(dt, g["shape"], g["seed_" + dt])
    vol = wide_image(dt, g["vol_shape"], int(g["seed_" + dt]) + 100)
    xc, yc, fact, coef = g["xcenter"], g["ycenter"], g["list_fact"], g["list_coef"]
    imd = im.astype(np.float64)          # what scipy reads: every element as a double
    for order in (0, 1, 3):
        pre_r = orc.unwarp_image_backward(imd, xc, yc, fact, order=order) if order > 1 else None
        pre_p = orc.map_coordinates(imd, g["pts_y"], g["pts_x"], order) if order > 1 else None
        assert wide_close(orc.unwarp_image_backward(im, xc, yc, fact, order=order), g["radial_o%d_%s" % (order, dt)], order, pre_r), order
        assert wide_close(orc.map_coordinates(im, g["pts_y"], g["pts_x"], order), g["points_o%d_%s" % (order, dt)], order, pre_p), order
    assert wide_close(orc.correct_perspective_image(im, coef), g["persp_o1_" + dt], 1)
    sl = orc.unwarp_slice_backward(vol, xc, yc, fact, int(g["index"]))
    assert sl.dtype == np.float32 and np.array_equal(sl, g["slice_" + dt])
    assert wide_close(orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, int(g["start"]), int(g["stop"])), g["chunk_" + dt], 1)
    if dt == "int64":       # the extremes really are in play: INT64_MAX reads as 2^63 and comes back as the x86 "integer indefinite"
        assert np.any(im == np.iinfo(np.int64).max) and not np.any(g["radial_o0_int64"] == np.iinfo(np.int64).max)
    if dt == "uint64":      # ... and UINT64_MAX reads as 2^64 and comes back as 0
        assert np.any(im == np.iinfo(np.uint64).max) and not np.any(g["radial_o0_uint64"] == np.iinfo(np.uint64).max)


def test_g12b_complex_is_two_real_interpolations(orc):
    g = golden("g12b_wide_types40x52")
    cim = (np.random.default_rng(861).random((40, 52)) + 1j * np.random.default_rng(862).random((40, 52))).astype(np.complex64)
    a = (g["xcenter"], g["ycenter"], g["list_fact"])
    want = g["radial_o1_complex64"]
    assert want.dtype == np.complex64
    assert np.array_equal(orc.unwarp_image_backward(np.ascontiguousarray(cim.real), *a), want.real)
    assert np.array_equal(orc.unwarp_image_backward(np.ascontiguousarray(cim.imag), *a), want.imag)


def test_g19_perspective_lines(orc):
    g = golden("g19_perspective_lines")
    lines = [np.array(line) for line in g["lines"]]
    fwd = orc.correct_perspective_line(lines, g["fcoef"])
    assert np.array_equal(np.asarray(fwd), g["forward"]) and np.array_equal(np.asarray(orc.correct_perspective_line(fwd, g["bcoef"])), g["back"])
