"""GPU suite (-m gpu): the HIP kernels, called through the C ABI by the reference-shaped Python
front end, against (a) the reference's own outputs in tests/golden/, (b) the CPU oracle on seeded
inputs, and (c) size-independent properties at the BASELINE sizes.

Bar: bit-exact for order 0 and for blend="scipy"; the default blend ("f64lerp") may differ from
scipy's arithmetic by at most one float32 ulp (and in these vectors does not differ at all).
"""
import numpy as np
import pytest

from conftest import DEV, HOST, G12B_DTYPES, G12_DTYPES, g12_inputs, golden, noise, typed_image, ulp_diff, wide_image

pytestmark = pytest.mark.gpu

from discorpy_amd import configs  # noqa: E402
from discorpy_amd.post import postprocessing as pp  # noqa: E402


def kernel_oracle(orc, blend):
    return dict(poly=orc.POLY_KERNEL, blend={"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP,
                                              "f32": orc.BLEND_F32LERP}[blend])


# --------------------------------------------------------------------------- (a) golden vectors

@pytest.mark.parametrize("blend", ["scipy", "f64lerp"])
def test_g1_reference_box_image(hip, blend):
    g = golden("g1_box64")
    a = (g["mat"], float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    assert np.array_equal(pp.unwarp_image_backward(*a, blend=blend), g["out_order1"])
    assert np.array_equal(pp.unwarp_image_backward(*a, order=0), g["out_order0"])
    # the reference's own assertion (tests/test_postprocessing.py:83-85)
    vals = np.mean(pp.unwarp_image_backward(*a), axis=0)[11:-10]
    pos = len(vals) // 2
    assert vals[0] < vals[pos] and vals[-1] < vals[pos]


@pytest.mark.parametrize("blend", ["scipy", "f64lerp"])
def test_g2_reference_slice_and_chunk(hip, blend):
    g = golden("g2_stripes10x64x64")
    vol = np.repeat(g["mat"][None], int(g["depth"]), axis=0)
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    s = pp.unwarp_slice_backward(vol, *a, int(g["index"]), blend=blend)
    assert s.dtype == np.float32 and s.shape == (10, 64) and np.array_equal(s, g["slice_out"])
    c = pp.unwarp_chunk_slices_backward(vol, *a, int(g["start"]), int(g["stop"]), blend=blend)
    assert c.shape == (10, 11, 64) and np.array_equal(c, g["chunk_out"])
    # the reference's own assertions (tests/test_postprocessing.py:107-108, 121-123)
    y0 = int(g["index"])
    assert np.max(vol[:, y0, :] - s) > 0.1
    assert np.max(vol[:, y0 - 5, :] - c[:, 0, :]) > 0.1 and np.max(vol[:, y0 + 5, :] - c[:, -1, :]) > 0.1


@pytest.mark.parametrize("blend", ["scipy", "f64lerp"])
def test_g3_reference_perspective(hip, blend):
    g = golden("g3_perspective64")
    c1 = pp.correct_perspective_image(g["mat"], list(g["coef_backward"]), blend=blend)
    assert np.array_equal(c1, g["cor_backward"])
    c2 = pp.correct_perspective_image(c1, list(g["coef_forward"]), blend=blend)
    assert np.array_equal(c2, g["cor_forward_of_backward"])
    assert np.array_equal(pp.correct_perspective_image(g["mat"], list(g["coef_backward"]), order=0),
                          g["cor_backward_order0"])
    # the reference's own assertions (tests/test_postprocessing.py:226-238)
    l0, l1, l2 = np.mean(g["mat"], axis=1), np.mean(c1, axis=1), np.mean(c2, axis=1)
    assert len(l1[l1 > 0]) > len(l0[l0 > 0]) and np.argmax(l0) == np.argmax(l2)


@pytest.mark.parametrize("blend", ["scipy", "f64lerp"])
def test_g4_config1_dot_pattern_05(hip, blend):
    g = golden("g4_dot_pattern_05")
    out = pp.unwarp_image_backward(g["crop_in"], float(g["crop_xcenter"]), float(g["crop_ycenter"]),
                                   list(g["list_fact"]), blend=blend)
    assert np.array_equal(out, g["crop_out"])


@pytest.mark.parametrize("name", ["g5_cfg2_160", "g5_offcentre_150x200", "g5_cfg5_9term_144"])
def test_g5_configs_reduced(hip, name):
    g = golden(name)
    img = noise(g["seed"], g["shape"])
    a = (img, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    assert np.array_equal(pp.unwarp_image_backward(*a, blend="scipy"), g["out_order1"])
    assert np.array_equal(pp.unwarp_image_backward(*a, order=0), g["out_order0"])
    assert ulp_diff(pp.unwarp_image_backward(*a), g["out_order1"]).max() <= 1
    assert np.max(np.abs(pp.unwarp_image_backward(*a, blend="f32").astype(np.float64) - g["out_order1"])) <= 1e-5


def test_g6_stack_rows(hip):
    g = golden("g6_stack3x800x1280")
    vol = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    for r in g["rows"]:
        assert np.array_equal(pp.unwarp_slice_backward(vol, *a, int(r), blend="scipy"), g["slice_%d" % r])
    assert np.array_equal(pp.unwarp_slice_backward(vol, *a, 400.5, blend="scipy"), g["slice_frac_400p5"])
    for key, (s0, s1) in {"chunk_395_402": (395, 402), "chunk_0_2": (0, 2), "chunk_797_799": (797, 799)}.items():
        assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, s0, s1, blend="scipy"), g[key])
        assert ulp_diff(pp.unwarp_chunk_slices_backward(vol, *a, s0, s1), g[key]).max() <= 1


def test_g15_folding_model_reflects_inside_the_reference_band(hip, orc):
    """See tests/test_oracle_golden.py: under a folding model the chunk function reflects row coordinates inside the band it
    crops; the HIP path (the direct stack kernel with the per-pixel band check) equals the reference on every pixel."""
    g = golden("g15_folding_chunk")
    vol = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]), int(g["start"]), int(g["stop"]))
    out = pp.unwarp_chunk_slices_backward(vol, *a, blend="scipy")
    assert hip.last_kernel().startswith("stack_rows_kernel"), hip.last_kernel()
    assert np.array_equal(out, g["ref_out"])
    assert ulp_diff(pp.unwarp_chunk_slices_backward(vol, *a), g["ref_out"])[:, ~g["outside_band"]].max() <= 1
    assert np.array_equal(pp.unwarp_chunk_slices_backward(vol.astype(np.float64), *a), g["ref_out_f64"])
    assert np.array_equal(pp.unwarp_chunk_slices_backward((vol * 60000).astype(np.uint16), *a), g["ref_out_u16"])
    vol2 = noise(g["case2_seed"], g["case2_shape"])
    a2 = (float(g["case2_xcenter"]), float(g["case2_ycenter"]), list(g["case2_list_fact"]), int(g["case2_rows"][0]), int(g["case2_rows"][1]))
    assert np.array_equal(pp.unwarp_chunk_slices_backward(vol2, *a2, blend="scipy"), g["case2_ref_out"])
    assert np.array_equal(pp.unwarp_chunk_slices_backward(vol2.astype(np.float64), *a2), g["case2_ref_out_f64"])
    # a sub-chunk has its own band, as a call of the reference on those rows would
    sub = (a[0], a[1], a[2], a[3] + 3, a[4] - 2)
    assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *sub, blend="scipy"), orc.unwarp_chunk_slices_backward(vol, *sub))


def test_g16_map_index_outside_the_image_follows_the_reference(hip):
    """correct_perspective_image(map_index=..., mode=...) with coordinates outside the image: the reference's outputs
    (it hands both to scipy), every mode, orders 0 and 1."""
    g = golden("g16_map_index_outside")
    mat = noise(g["seed"], g["shape"])
    coef = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        for order in (0, 1):
            got = pp.correct_perspective_image(mat, coef, order=order, mode=mode, map_index=(g["ys"], g["xs"]), blend="scipy")
            assert np.array_equal(got, g["%s_o%d" % (mode.replace("-", "_"), order)]), (mode, order)


def test_g17_small_frames_every_order_and_mode_equal_the_reference(hip):
    """Golden G17 through the HIP path: bit-equal to the reference's outputs (scipy's exact blend at order 1)."""
    from test_oracle_golden import g17_cases
    g = golden("g17_small_frames_orders_modes")
    for k, img, xc, yc, fact, order, mode, coef in g17_cases():
        assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode, blend="scipy"), g["radial_%02d" % k]), (k, order, mode)
        assert np.array_equal(pp.correct_perspective_image(img, coef, order=order, mode=mode, blend="scipy"), g["persp_%02d" % k]), (k, order, mode)


def test_g18_slice_with_any_index_equals_the_reference(hip):
    from test_oracle_golden import g18_cases
    g = golden("g18_slice_any_index")
    for t, vol, xc, yc, fact, idx in g18_cases():
        out = pp.unwarp_slice_backward(vol, xc, yc, fact, idx, blend="scipy")
        assert out.dtype == np.float32 and np.array_equal(out, g["slice_%02d" % t]), (t, vol.dtype, idx)


def test_g7_fused_and_two_pass(hip):
    g = golden("g7_fused144")
    img = noise(g["seed"], g["shape"])
    r = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    coef = list(g["list_coef"])
    assert np.array_equal(pp.unwarp_perspective_fused(img, *r, coef, blend="scipy"), g["fused_out"])
    two = pp.correct_perspective_image(pp.unwarp_image_backward(img, *r, blend="scipy"), coef, blend="scipy")
    assert np.array_equal(two, g["twopass_out"])
    assert np.array_equal(pp.correct_perspective_image(img, coef, blend="scipy"), g["persp_out"])
    # map_index= path with the reference's own float32 coordinate planes
    via_map = pp.correct_perspective_image(img, coef, map_index=(g["yd"].reshape(-1, 1), g["xd"].reshape(-1, 1)),
                                           blend="scipy")
    assert np.array_equal(via_map, g["fused_out"])


def test_coordinate_planes_equal_the_references(hip, orc):
    """The float32 (yd, xd) planes computed on the GPU against the planes numpy produced in the
    reference (postprocessing.py:141-145, 444-459): the sharpest statement of coordinate parity."""
    for name in ("g5_cfg2_160", "g5_offcentre_150x200", "g5_cfg5_9term_144", "g7_fused144"):
        g = golden(name)
        shape = tuple(int(v) for v in g["shape"])
        if name == "g7_fused144":
            yd, xd = pp.generate_fused_map(shape, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]),
                                           list(g["list_coef"]))
        else:
            yd, xd = pp.generate_radial_map(shape, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
        assert yd.dtype == np.float32 and np.array_equal(yd, g["yd"]) and np.array_equal(xd, g["xd"]), name
    # _generate_perspective_map: same call, same return shape as the reference; feeding it back as
    # map_index reproduces correct_perspective_image
    g = golden("g7_fused144")
    img = noise(g["seed"], g["shape"])
    ymap, xmap = pp._generate_perspective_map(img, list(g["list_coef"]))
    assert ymap.shape == (img.size, 1) and xmap.shape == (img.size, 1) and ymap.dtype == np.float32
    oy, ox = orc.perspective_coords(img.shape[0], img.shape[1], list(g["list_coef"]))
    assert np.array_equal(ymap.reshape(img.shape), oy.astype(np.float32))
    assert np.array_equal(xmap.reshape(img.shape), ox.astype(np.float32))
    assert np.array_equal(pp.correct_perspective_image(img, list(g["list_coef"]), map_index=(ymap, xmap), blend="scipy"),
                          g["persp_out"])
    # full cfg2 frame: every one of the 33.5 M coordinates against the oracle's kernel order
    c = configs.cfg2()
    yd, xd = pp.generate_radial_map(c["shape"], c["xcenter"], c["ycenter"], c["list_fact"])
    oy, ox = orc.radial_coords(c["shape"][0], c["shape"][1], c["xcenter"], c["ycenter"], c["list_fact"],
                               poly=orc.POLY_KERNEL)
    assert np.array_equal(yd, oy.astype(np.float32)) and np.array_equal(xd, ox.astype(np.float32))
    # ... and against numpy's own evaluation order: 0-2 rounding-boundary coordinates per frame
    ny, nx = orc.radial_coords(c["shape"][0], c["shape"][1], c["xcenter"], c["ycenter"], c["list_fact"],
                               poly=orc.POLY_NUMPY)
    flips = int((yd != ny.astype(np.float32)).sum() + (xd != nx.astype(np.float32)).sum())
    assert flips <= 4, flips


def test_g8_clipping_stress_all_modes(hip):
    g = golden("g8_clip120x180")
    img = noise(g["seed"], g["shape"])
    a = (img, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        assert np.array_equal(pp.unwarp_image_backward(*a, mode=mode, blend="scipy"), g["out_order1"])
        assert np.array_equal(pp.unwarp_image_backward(*a, order=0, mode=mode), g["out_order0"])


def test_g9_explicit_coordinates(hip):
    g = golden("g9_points33x47")
    img = noise(g["seed"], g["shape"])
    assert np.array_equal(pp.remap_coordinates(img, g["ys"], g["xs"], order=0), g["out_order0"])
    assert np.array_equal(pp.remap_coordinates(img, g["ys"], g["xs"], blend="scipy"), g["out_order1"])
    assert np.array_equal(pp.remap_coordinates(img, g["ys64"], g["xs64"], order=0), g["out64_order0"])
    assert np.array_equal(pp.remap_coordinates(img, g["ys64"], g["xs64"], blend="scipy"), g["out64_order1"])


MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")


def spline_close(got, ref):
    d = ulp_diff(got, ref)
    return d.max() <= 1 and np.count_nonzero(d) <= max(2, got.size // 500)


@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_g11_spline_orders_against_the_reference(hip, orc, order):
    """Orders 2..5 (scipy's prefiltered B-splines): bit-equal to the reference in every boundary mode (since round 2: the
    prefilter reproduces scipy's boundary initialisations to the last detail), and to the oracle."""
    g = golden("g11_spline45x60")
    img = noise(g["seed"], g["shape"])
    a = (img, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    for mode in MODES:
        out = pp.unwarp_image_backward(*a, order=order, mode=mode)
        assert np.array_equal(out, g["radial_o%d_%s" % (order, mode)]), (order, mode)
        assert np.array_equal(out, orc.unwarp_image_backward(*a, order=order, mode=mode, poly=orc.POLY_KERNEL)), (order, mode)
    pts = pp.remap_coordinates(img, g["pts_y"], g["pts_x"], order=order)
    assert np.array_equal(pts, g["points_o%d_reflect" % order])
    assert np.array_equal(pts, orc.remap_coords(img, g["pts_y"], g["pts_x"], order=order))


def test_g11_perspective_order3_as_demo_07(hip, orc):
    g = golden("g11_spline45x60")
    img = noise(g["seed"], g["shape"])
    coef = list(g["list_coef"])
    for mode in MODES:
        out = pp.correct_perspective_image(img, coef, order=3, mode=mode)
        assert np.array_equal(out, g["persp_o3_%s" % mode]), mode
        assert np.array_equal(out, orc.correct_perspective_image(img, coef, order=3, mode=mode)), mode
    ymap, xmap = pp._generate_perspective_map(img, coef)
    assert np.array_equal(pp.correct_perspective_image(img, coef, order=3, map_index=(ymap, xmap)),
                          pp.correct_perspective_image(img, coef, order=3))


def test_g13_unwarp_line_forward(hip):
    """Closed-form point mapping on the GPU (dcp_map_points_f64) against the reference's numpy/libm evaluation."""
    g = golden("g13_lines_forward")
    lines = [np.array(line) for line in g["lines"]]
    out = pp.unwarp_line_forward(lines, float(g["xcenter"]), float(g["ycenter"]), g["list_fact"])
    assert len(out) == 8 and all(o.shape == (12, 2) and o.dtype == np.float64 for o in out)
    assert np.allclose(np.asarray(out), g["out"], rtol=1e-13, atol=1e-10)
    assert pp.unwarp_line_forward([], 1.0, 2.0, [1.0]) == []
    ints = pp.unwarp_line_forward([np.array([[10, 20], [30, 40]])], 25.0, 25.0, [1.0, 1e-3])   # integer lines stay integer
    assert ints[0].dtype.kind == "i"


def test_g19_correct_perspective_line(hip, orc):
    """The homography on point lists (dcp_map_points_perspective_f64; reference postprocessing.py:414-441): numpy's operation order
    with IEEE divisions, so bit-equal to the reference's own outputs on the reference's own test lines."""
    g = golden("g19_perspective_lines")
    lines = [np.array(line) for line in g["lines"]]
    fwd = pp.correct_perspective_line(lines, list(g["fcoef"]))
    assert len(fwd) == 4 and all(o.shape == (32, 2) and o.dtype == np.float64 for o in fwd)
    assert np.array_equal(np.asarray(fwd), g["forward"])
    assert np.array_equal(np.asarray(pp.correct_perspective_line(fwd, list(g["bcoef"]))), g["back"])
    assert np.array_equal(np.asarray(orc.correct_perspective_line(lines, g["fcoef"])), g["forward"])
    with pytest.raises(ValueError, match="Eight coefficients"):
        pp.correct_perspective_line(lines, [1.0] * 7)
    assert pp.correct_perspective_line([], list(g["fcoef"])) == []
    rng = np.random.default_rng(3)
    big = [rng.uniform(-50, 4000, (n, 2)) for n in (1, 1000, 0, 37)]
    coef = [0.97, 0.02, 15.0, -0.01, 0.98, 8.0, 1e-5, -2e-5]
    got, want = pp.correct_perspective_line(big, coef), orc.correct_perspective_line([b for b in big if len(b)], coef)
    assert got[2].size == 0 and all(np.array_equal(a, b) for a, b in zip([q for q in got if q.size], want))


def test_fused_map_at_spline_orders(hip, orc):
    """The one-pass perspective -> radial remap at orders 2..5: equal to sampling the image at the fused coordinate
    planes (which another test holds bit-equal to the reference's numpy planes), float32 and uint16."""
    a = (61.0, 47.0, [1.0, 2e-3, -1e-6])
    coef = [0.97, 0.02, 1.5, -0.01, 0.98, 0.8, 1e-4, -2e-4]
    for img in (noise(33, (90, 120)), typed_image("uint16", (90, 120), 34)):
        py, px = pp.generate_fused_map(img.shape, *a, coef)
        for order, mode in ((2, "reflect"), (3, "mirror"), (3, "nearest"), (5, "grid-wrap")):
            got = pp.unwarp_perspective_fused(img, *a, coef, order=order, mode=mode)
            want = orc.map_coordinates(img, py, px, order, mode)
            assert got.dtype == img.dtype
            if img.dtype == np.float32:
                assert spline_close(got, want), (order, mode)
            else:
                assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1, (order, mode)


def test_spline_orders_on_ragged_and_large_inputs(hip, orc):
    for shape in [(2, 2), (3, 17), (40, 1), (129, 300)]:
        img = (noise(shape[1], shape) * 100).astype(np.float32)
        for order, mode in [(3, "reflect"), (2, "mirror"), (5, "nearest"), (4, "grid-wrap"), (3, "constant")]:
            want = orc.unwarp_image_backward(img, 0.4 * shape[1], 0.6 * shape[0], [1.0, 2e-3], order=order, mode=mode,
                                             poly=orc.POLY_KERNEL)
            got = pp.unwarp_image_backward(img, 0.4 * shape[1], 0.6 * shape[0], [1.0, 2e-3], order=order, mode=mode)
            assert np.array_equal(got, want), (shape, order, mode)
    rgb = noise(9, (50, 70, 3))                                   # strided channel view, as demo_07 loops
    for ch in range(3):
        want = orc.correct_perspective_image(np.ascontiguousarray(rgb[:, :, ch]), [0.98, 0.01, 1.0, -0.01, 0.97, 2.0, 1e-4, -1e-4],
                                             order=3)
        assert np.array_equal(pp.correct_perspective_image(rgb[:, :, ch], [0.98, 0.01, 1.0, -0.01, 0.97, 2.0, 1e-4, -1e-4], order=3), want)
    # lines longer than 256 samples are prefiltered in overlapping chunks on the GPU: the float64
    # coefficients then agree with the serial recursion to ~1e-24 relative, not bit for bit
    c = configs.cfg2()
    img = noise(c["seed"], (1024, 4096))
    for order, mode in [(3, "reflect"), (5, "mirror"), (2, "grid-wrap"), (4, "nearest")]:
        got = pp.unwarp_image_backward(img, c["xcenter"], 500.0, c["list_fact"], order=order, mode=mode)
        want = orc.unwarp_image_backward(img, c["xcenter"], 500.0, c["list_fact"], order=order, mode=mode,
                                         poly=orc.POLY_KERNEL)
        assert spline_close(got, want), (order, mode)
        assert np.count_nonzero(got != want) <= 8, (order, mode)


def test_spline_one_pass_prefilter_and_lds_gather_against_the_plain_kernels(hip, orc):
    """Frames large enough for the one-pass tile prefilter (spline_tile_filter_kernel: reflect / mirror kinds, z^n == 0)
    and the LDS-staged gather (spline_wg_kernel: certified radial / perspective maps): equal to the chunked passes +
    global gather (options spline_tiled = spline_wg = 0) and to the oracle up to the restart error of long lines."""
    F = hip
    c = configs.cfg2()
    shape = (1100, 1347)          # partial tiles on both axes
    img = noise(31, shape)
    coef = [1.02, 0.015, -9.0, -0.012, 0.99, 6.0, 2.0e-6, -1.5e-6]
    try:
        for order, mode in [(2, "mirror"), (3, "reflect"), (3, "nearest"), (4, "reflect"), (5, "grid-constant"), (5, "reflect")]:
            a = (img, c["xcenter"] * shape[1] / 4096.0, 500.0, c["list_fact"])
            res = {}
            # 1: the default (one-pole orders: the register column pass for float32 frames that need no padding, the register row
            # pass behind an odd-pitch LDS staging); 2: the LDS tile kernel on both axes; 0: the plain kernels
            # 1: the default (one-pole orders: both axes in one launch for float32 frames that need no padding); 6: the two launches of
            # rounds 3-5 (the register column pass, the register row pass behind an odd-pitch LDS staging); 2: the LDS tile kernel on both
            # axes; 0: the plain kernels
            for fast in (1, 6, 2, 0):
                F.set_option("x_spline_tiled", fast)
                F.set_option("x_spline_wg", 1 if fast else 0)
                res[fast] = (pp.unwarp_image_backward(*a, order=order, mode=mode),
                             pp.correct_perspective_image(img, coef, order=order, mode=mode))
                one_pole = fast in (1, 6) and order <= 3
                direct = one_pole and mode not in ("nearest", "grid-constant")        # (those two pad the plane first)
                pre = ("spline_prefilter2d_kernel" if one_pole and fast == 1 else       # (round 5: the padded modes too)
                       "spline_prefilter2d_kernel x 2" if fast == 1 else                # (round 6: the two-pole orders, one pass per pole)
                       "spline_col_lds_kernel + spline_row_lds_kernel" if direct else
                       "spline_tile_filter_kernel + spline_row_lds_kernel" if one_pole else "spline_tile_filter_kernel x 2")
                want_name = (pre + " + spline_wg_kernel<order=%d>" if fast else
                             "spline_causal / anticausal / transpose kernels + spline_remap_kernel<order=%d>") % order
                assert F.last_kernel() == want_name, F.last_kernel()
            want = (orc.unwarp_image_backward(*a, order=order, mode=mode, poly=orc.POLY_KERNEL),
                    orc.correct_perspective_image(img, coef, order=order, mode=mode))
            for k in (0, 1):
                assert np.count_nonzero(res[1][k] != res[0][k]) <= 4 and np.count_nonzero(res[2][k] != res[0][k]) <= 4, (order, mode, k)
                assert np.count_nonzero(res[1][k] != res[2][k]) <= 4 and np.count_nonzero(res[1][k] != res[6][k]) <= 4, (order, mode, k)
                assert spline_close(res[1][k], want[k]) and np.count_nonzero(res[1][k] != want[k]) <= 8, (order, mode, k)
    finally:
        F.set_option("x_spline_tiled", 1)
        F.set_option("x_spline_wg", 1)


# --------------------------------------------------------------------------- (b) oracle, seeded inputs

SHAPES = [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (16, 64), (17, 65), (63, 257), (300, 517), (129, 1031)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("blend", ["scipy", "f64lerp", "f32"])
def test_radial_matches_oracle_on_ragged_shapes(hip, orc, shape, blend):
    img = (noise(hash(shape) % 1000, shape) * 255).astype(np.float32)
    xc, yc = 0.43 * shape[1], 0.61 * shape[0]
    fact = [1.01, -4e-4, 3e-7]
    want = orc.unwarp_image_backward(img, xc, yc, fact, **kernel_oracle(orc, blend))
    assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact, blend=blend), want)
    want0 = orc.unwarp_image_backward(img, xc, yc, fact, order=0, poly=orc.POLY_KERNEL)
    assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact, order=0), want0)


@pytest.mark.parametrize("nfact", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 17, 32])
def test_every_polynomial_length(hip, orc, nfact):
    """1..10 terms run the SGPR-resident unrolled polynomial, the rest the LDS-staged one."""
    img = noise(nfact, (90, 140))
    rng = np.random.default_rng(100 + nfact)
    fact = [1.0] + [float(rng.uniform(-1, 1)) * 10.0 ** (-2.2 * i) for i in range(1, nfact)]
    fact = fact[:nfact]
    want = orc.unwarp_image_backward(img, 70.3, 44.9, fact, **kernel_oracle(orc, "scipy"))
    assert np.array_equal(pp.unwarp_image_backward(img, 70.3, 44.9, fact, blend="scipy"), want)
    hip.set_option("x_coef_lds", 1)
    try:
        assert np.array_equal(pp.unwarp_image_backward(img, 70.3, 44.9, fact, blend="scipy"), want)
    finally:
        hip.set_option("x_coef_lds", 0)
    with pytest.raises(ValueError):
        pp.unwarp_image_backward(img, 1, 1, [1.0] * 33)


def test_centre_on_a_pixel_and_far_outside(hip, orc):
    img = noise(9, (64, 96))
    for xc, yc in [(32.0, 16.0), (0.0, 0.0), (-500.0, 40.0), (2000.0, -3000.0), (95.0, 63.0)]:
        for fact in ([1.0, 3e-3], [0.0, 1e-2], [1.3, 2e-3, -1e-5]):
            want = orc.unwarp_image_backward(img, xc, yc, fact, **kernel_oracle(orc, "scipy"))
            assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact, blend="scipy"), want)


def test_tuning_knobs_do_not_change_results(hip, orc):
    img = noise(4, (200, 1700))          # 27 tile columns: uneven XCD stripes
    a = (img, 833.3, 80.8, list(configs.COEF_DOT_05))
    want = orc.unwarp_image_backward(*a, **kernel_oracle(orc, HOST))
    keys = {"x_tile_rows": [1, 3, 8, 16, 64], "x_pipe_depth": [1, 2, 4], "x_xcd_remap": [0, 1, 2], "lds_gather": [0, 1]}
    for key, vals in keys.items():
        old = hip.get_option(key)
        try:
            for v in vals:
                hip.set_option(key, v)
                assert np.array_equal(pp.unwarp_image_backward(*a), want), (key, v)
        finally:
            hip.set_option(key, old)


def test_lds_staged_gather_and_its_fallbacks(hip, orc):
    """remap_lds_kernel: staged tiles, tiles whose source box does not fit (strong minification),
    partial tiles at the right and bottom edges -- all bit-equal to the direct gather."""
    old = hip.get_option("lds_gather")
    hip.set_option("lds_gather", 1)
    try:
        hip.debug_counters()
        for shape, xc, yc, fact in [((130, 200), 99.0, 61.0, [1.0, 1e-4]),         # gentle: all staged
                                    ((97, 333), 150.0, 40.0, [1.0, 2e-3]),         # strong barrel
                                    ((64, 64), 31.5, 31.5, [2.5]),                 # 2.5x minification: box too large
                                    ((100, 100), 50.0, 50.0, [0.2, 1e-2])]:
            img = noise(shape[0], shape)
            for blend in ("scipy", "f64lerp", "f32"):
                want = orc.unwarp_image_backward(img, xc, yc, fact, **kernel_oracle(orc, blend))
                assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact, blend=blend), want)
        nofit, vote = hip.debug_counters()
        assert nofit > 0            # the minification case must have exercised the fallback
    finally:
        hip.set_option("lds_gather", old)


def test_strided_and_padded_sources(hip, orc):
    rgb = noise(7, (120, 150, 3))
    a = (74.0, 59.5, [1.0, 1e-3, 2e-6])
    for ch in range(3):
        view = rgb[:, :, ch]                                    # column stride 3 (demo_06.py:111-113)
        want = orc.unwarp_image_backward(np.ascontiguousarray(view), *a, **kernel_oracle(orc, "scipy"))
        assert np.array_equal(pp.unwarp_image_backward(view, *a, blend="scipy"), want)
    padded = np.zeros((120, 256), np.float32)
    padded[:, :150] = rgb[:, :, 0]
    want = orc.unwarp_image_backward(np.ascontiguousarray(rgb[:, :, 0]), *a, **kernel_oracle(orc, "scipy"))
    assert np.array_equal(pp.unwarp_image_backward(padded[:, :150], *a, blend="scipy"), want)
    assert np.array_equal(pp.unwarp_image_backward(rgb[::-1, :, 0], *a, blend="scipy"),
                          orc.unwarp_image_backward(np.ascontiguousarray(rgb[::-1, :, 0]), *a,
                                                    **kernel_oracle(orc, "scipy")))
    band = noise(8, (5, 90, 130))[2, 10:70, :]                  # row-band view of a projection
    assert np.array_equal(pp.unwarp_image_backward(band, 60.0, 30.0, [1.0, 1e-3], blend="scipy"),
                          orc.unwarp_image_backward(np.ascontiguousarray(band), 60.0, 30.0, [1.0, 1e-3],
                                                    **kernel_oracle(orc, "scipy")))


@pytest.mark.parametrize("shape", [(2, 2), (33, 47), (128, 300), (257, 130)])
def test_perspective_and_fused_match_oracle(hip, orc, shape):
    img = noise(shape[1], shape)
    h, w = shape
    coef = [0.97, -0.02, 0.03 * w, 0.015, 0.95, 0.02 * h, -2e-5 * 64 / w, 3e-5 * 64 / h]
    for blend in ("scipy", "f64lerp", "f32"):
        ob = kernel_oracle(orc, blend)["blend"]
        assert np.array_equal(pp.correct_perspective_image(img, coef, blend=blend),
                              orc.correct_perspective_image(img, coef, blend=ob))
        for fact in ([1.0, 1e-3], list(configs.COEF_DOT_05), [1.0, -1e-3, 2e-6, 1e-9, 1e-12, 1e-15]):
            assert np.array_equal(pp.unwarp_perspective_fused(img, 0.45 * w, 0.55 * h, fact, coef, blend=blend),
                                  orc.unwarp_fused(img, 0.45 * w, 0.55 * h, fact, coef, poly=orc.POLY_KERNEL, blend=ob))
    assert np.array_equal(pp.correct_perspective_image(img, coef, order=0),
                          orc.correct_perspective_image(img, coef, order=0))
    assert np.array_equal(pp.unwarp_perspective_fused(img, 0.45 * w, 0.55 * h, [1.0, 1e-3], coef, order=0),
                          orc.unwarp_fused(img, 0.45 * w, 0.55 * h, [1.0, 1e-3], coef, order=0, poly=orc.POLY_KERNEL))


def test_wild_homographies_take_the_full_division(hip, orc):
    """A denominator that changes sign inside the image, or huge coefficients, disable the
    shared-reciprocal division on the host side; results must still equal IEEE division."""
    img = noise(21, (96, 130))
    h, w = img.shape
    wild = [[1.0, 0.0, 0.0, 0.0, 1.0, 0.0, -1.0 / (w / 2 + 0.37), 0.0],          # pole inside the image
            [1.0, 0.1, 3.0, 0.05, 1.0, 2.0, 0.0, -1.0 / (h / 3 + 0.21)],
            [1e150, 0.0, 0.0, 0.0, 1e-150, 0.0, 0.0, 0.0],                         # out-of-range magnitudes
            [0.9, 0.0, 4.0, 0.0, 1.1, -3.0, 1e-3, 2e-3]]                           # tame control
    for coef in wild:
        for blend in ("scipy", "f64lerp"):
            ob = kernel_oracle(orc, blend)["blend"]
            assert np.array_equal(pp.correct_perspective_image(img, coef, blend=blend),
                                  orc.correct_perspective_image(img, coef, blend=ob))
            assert np.array_equal(pp.unwarp_perspective_fused(img, 60.0, 50.0, [1.0, 1e-3], coef, blend=blend),
                                  orc.unwarp_fused(img, 60.0, 50.0, [1.0, 1e-3], coef, poly=orc.POLY_KERNEL, blend=ob))
        assert np.array_equal(pp.correct_perspective_image(img, coef, order=0),
                              orc.correct_perspective_image(img, coef, order=0))


def test_stack_rows_match_oracle(hip, orc):
    vol = noise(13, (7, 120, 200))
    a = (97.3, 66.1, [1.004, -6e-5, 3e-7])
    for index in (0, 1, 59, 60.25, 119, -3, 130):              # the reference does not validate `index`
        want = orc.unwarp_slice_backward(vol, *a, index, **kernel_oracle(orc, "scipy"))
        assert np.array_equal(pp.unwarp_slice_backward(vol, *a, index, blend="scipy"), want)
    for s0, s1 in [(0, 0), (0, 119), (40, 70), (118, 119)]:
        for blend in ("scipy", "f64lerp", "f32"):
            want = orc.unwarp_chunk_slices_backward(vol, *a, s0, s1, **kernel_oracle(orc, blend))
            assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, s0, s1, blend=blend), want)
    for key, vals in {"x_d_chunk": [1, 3, 64]}.items():
        old = hip.get_option(key)
        try:
            for v in vals:
                hip.set_option(key, v)
                assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, 40, 70, blend="scipy"),
                                      orc.unwarp_chunk_slices_backward(vol, *a, 40, 70, **kernel_oracle(orc, "scipy")))
        finally:
            hip.set_option(key, old)
    empty = pp.unwarp_slice_backward(np.zeros((0, 8, 8), np.float32), 4, 4, [1.0], 3)
    assert empty.shape == (0, 8)
    # non-contiguous stack (every other projection, a row band): handled by the front end
    assert np.array_equal(pp.unwarp_slice_backward(vol[::2], *a, 59, blend="scipy"),
                          orc.unwarp_slice_backward(np.ascontiguousarray(vol[::2]), *a, 59, **kernel_oracle(orc, "scipy")))


def test_lds_staged_stack_kernel_and_its_fallbacks(hip, orc):
    """stack_lds_kernel (chunks of rows, float32 coordinates) is chosen by launch size; forced here on small stacks:
    ragged tiles, every polynomial path, all blends, a strong model whose boxes do not fit (direct-gather fallback),
    odd depth chunks, padded projections."""
    torch = pytest.importorskip("torch")
    old = hip.get_option("x_stack_lds"), hip.get_option("x_d_chunk")
    try:
        for force in (2, 0):
            hip.set_option("x_stack_lds", force)
            for (d, h, w, r0, r1), fact in [((5, 100, 140, 20, 60), [1.0, 2e-3]), ((3, 77, 263, 0, 76), list(configs.COEF_DOT_05)),
                                            ((4, 90, 130, 10, 17), [1.0, 1e-3, 1e-6, 1e-9, 1e-12, 1e-15]),
                                            ((2, 200, 300, 50, 150), [0.4, 8e-3]), ((7, 64, 65, 3, 40), [1.0, -2e-3, 3e-5])]:
                vol = noise(500 + d, (d, h, w))
                a = (0.53 * w, 0.48 * h, fact)
                for blend in ("scipy", "f64lerp", "f32"):
                    want = orc.unwarp_chunk_slices_backward(vol, *a, r0, r1, **kernel_oracle(orc, blend))
                    for dc in (1, 3, 16):
                        hip.set_option("x_d_chunk", dc)
                        got = pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), *a, r0, r1, blend=blend)
                        assert np.array_equal(got.cpu().numpy(), want), (force, d, h, w, blend, dc)
                big = np.zeros((d, h + 4, w + 9), np.float32)
                big[:, 2:h + 2, 5:w + 5] = vol
                view = torch.from_numpy(big).cuda()[:, 2:h + 2, 5:w + 5]
                assert np.array_equal(pp.unwarp_chunk_slices_backward(view, *a, r0, r1).cpu().numpy(),
                                      orc.unwarp_chunk_slices_backward(vol, *a, r0, r1, **kernel_oracle(orc, "f64lerp")))
        hip.debug_counters()
        hip.set_option("x_stack_lds", 2)
        vol = noise(9, (2, 200, 300))
        pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), 150.0, 100.0, [0.4, 8e-3], 50, 150)
        nofit, vote = hip.debug_counters()
        assert nofit + vote > 0                      # that model really exercised the fallback
    finally:
        hip.set_option("x_stack_lds", old[0])
        hip.set_option("x_d_chunk", old[1])


def test_host_stack_sharded_over_devices_of_one_process(hip, orc):
    """dcp_unwarp_stack_rows_multi_f32: depth shards on one worker thread per entry of `devices` (here the one GPU of
    the box, several times) give the same sinograms as one call; ragged and empty shards included."""
    vol = noise(31, (7, 120, 160))
    a = (83.0, 55.0, list(configs.COEF_DOT_05))
    want = orc.unwarp_chunk_slices_backward(vol, *a, 30, 90, **kernel_oracle(orc, HOST))
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 9):
        assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, 30, 90, devices=devs), want), devs
    assert np.array_equal(pp.unwarp_slice_backward(vol, *a, 44, devices=[0, 0]),
                          orc.unwarp_slice_backward(vol, *a, 44, **kernel_oracle(orc, HOST)))
    with pytest.raises(ValueError, match="outside"):
        pp.unwarp_chunk_slices_backward(vol, *a, 30, 90, devices=[0, 99])
    with pytest.raises(ValueError, match="at least one device"):
        pp.unwarp_chunk_slices_backward(vol, *a, 30, 90, devices=[])


def test_host_stack_streams_in_depth_chunks(hip, orc):
    """DCP_MEM_HOST stacks go through the GPU chunk by chunk (upload k+1 while chunk k is copied back); the result
    must not depend on the chunking, including a ragged last chunk and chunks of a single projection."""
    vol = noise(91, (11, 90, 130))
    a = (66.0, 41.0, [1.0, 2e-3, 1e-6])
    want_c = orc.unwarp_chunk_slices_backward(vol, *a, 20, 70, **kernel_oracle(orc, HOST))
    want_s = orc.unwarp_slice_backward(vol, *a, 45, **kernel_oracle(orc, HOST))
    old = hip.get_option("stack_chunk_kb")
    try:
        for kb in (1, 64, 110, 24576):
            hip.set_option("stack_chunk_kb", kb)
            assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, 20, 70), want_c), kb
            assert np.array_equal(pp.unwarp_slice_backward(vol, *a, 45), want_s), kb
            padded = np.zeros((11, 95, 140), np.float32)
            padded[:, :90, :130] = vol
            assert np.array_equal(pp.unwarp_chunk_slices_backward(padded[:, :90, :130], *a, 20, 70), want_c), kb
            assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, 20, 70, devices=[0, 0, 0]), want_c), kb
    finally:
        hip.set_option("stack_chunk_kb", old)


class LazyStack:
    """Stands in for the h5py dataset losa.load_hdf_object returns: shape, dtype, slicing -- and a log of what was read."""

    def __init__(self, data):
        self._data = data
        self.shape = data.shape
        self.dtype = data.dtype
        self.reads = []

    def __getitem__(self, key):
        self.reads.append(key)
        return self._data[key]

    def __len__(self):
        return self.shape[0]


@pytest.mark.parametrize("dt", ["float32", "uint16"])
def test_out_of_core_stack_reads_only_the_row_band(hip, orc, dt, monkeypatch, tmp_path):
    """A dataset that is not an in-memory array is read like the reference reads it (:221-228): per depth chunk,
    only the rows the request can reach; results equal the in-memory path."""
    vol = typed_image(dt, (9, 300, 400), 12) if dt != "float32" else noise(12, (9, 300, 400))
    a = (190.0, 140.0, [1.0, 1.5e-3, 1e-6])
    want_c = pp.unwarp_chunk_slices_backward(vol, *a, 100, 139)
    want_s = pp.unwarp_slice_backward(vol, *a, 222)
    assert np.array_equal(want_c, orc.unwarp_chunk_slices_backward(vol, *a, 100, 139, poly=orc.POLY_KERNEL,
                                                                   **({"blend": kernel_oracle(orc, HOST)["blend"]} if dt == "float32" else {})))
    monkeypatch.setenv("DISCORPY_AMD_READ_CHUNK_MB", "0.2")
    lazy = LazyStack(vol)
    got = pp.unwarp_chunk_slices_backward(lazy, *a, 100, 139)
    assert got.dtype == vol.dtype and np.array_equal(got, want_c)
    b0, bn = hip.stack_row_band(300, 400, *a, 100, 40)
    assert 0 < b0 and bn < 80 and len(lazy.reads) >= 3
    covered = []
    for (ds, rs, cs) in lazy.reads:
        assert (rs.start, rs.stop) == (b0, b0 + bn) and cs == slice(None)
        covered += list(range(ds.start, ds.stop))
    assert covered == list(range(9))
    lazy = LazyStack(vol)
    sl = pp.unwarp_slice_backward(lazy, *a, 222)
    assert sl.dtype == np.float32 and np.array_equal(sl, want_s)
    sb0, sbn = hip.stack_row_band(300, 400, *a, 222, 1)
    assert sbn < 60 and all((r[1].start, r[1].stop) == (sb0, sb0 + sbn) for r in lazy.reads)
    # a memory-mapped .npy file is an ndarray: the library itself copies only the band
    path = tmp_path / "stack.npy"
    np.save(path, vol)
    mm = np.load(path, mmap_mode="r")
    assert np.array_equal(pp.unwarp_chunk_slices_backward(mm, *a, 100, 139), want_c)
    # the C ABI on a device-resident band, and its refusal of a band that is too small
    torch = pytest.importorskip("torch")
    if dt == "float32":
        L = hip.lib()
        fa, nf = hip.fact_array(a[2])
        band = torch.from_numpy(np.ascontiguousarray(vol[:, b0:b0 + bn, :])).cuda()
        out = torch.empty((9, 40, 400), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        hip.check(L.dcp_unwarp_stack_band(band.data_ptr(), out.data_ptr(), 0, 0, 9, 300, 400, b0, bn, bn * 400, 400, a[0], a[1],
                                          fa, nf, 100.0, 40, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE, 0, st))
        assert np.array_equal(out.cpu().numpy(), want_c)
        rc = L.dcp_unwarp_stack_band(band.data_ptr(), out.data_ptr(), 0, 0, 9, 300, 400, b0 + 3, bn - 3, bn * 400, 400, a[0],
                                     a[1], fa, nf, 100.0, 40, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE, 0, st)
        assert rc == hip.ERR_INVALID_ARG and "band holds" in hip.last_error()


def test_out_of_core_full_stack_correction(hip, orc, tmp_path, monkeypatch):
    """losa.stream.correct_stack: a stack on disk in, the corrected stack on disk out, pass by pass."""
    from discorpy_amd.losa import stream
    vol = typed_image("uint16", (7, 120, 200), 44)
    a = (96.0, 57.0, [1.0, 1.2e-3, 2e-6])
    np.save(tmp_path / "in.npy", vol)
    src = LazyStack(np.load(tmp_path / "in.npy", mmap_mode="r"))
    dst = np.lib.format.open_memmap(tmp_path / "out.npy", mode="w+", dtype=vol.dtype, shape=vol.shape)
    monkeypatch.setenv("DISCORPY_AMD_READ_CHUNK_MB", "0.05")
    assert stream.correct_stack(src, dst, *a, rows_per_pass=32) == 4
    dst.flush()
    got = np.load(tmp_path / "out.npy")
    want = np.stack([orc.unwarp_image_backward(vol[d], *a, poly=orc.POLY_KERNEL) for d in range(7)])
    assert got.dtype == vol.dtype and np.array_equal(got, want)          # corrected stack == every projection unwarped
    part = np.zeros((7, 30, 200), vol.dtype)
    assert stream.correct_stack(src, part, *a, row_range=(50, 80)) == 1 and np.array_equal(part, want[:, 50:80])
    with pytest.raises(ValueError, match="dst must have shape"):
        stream.correct_stack(src, np.zeros((7, 10, 200), vol.dtype), *a)


def test_host_frames_go_through_in_bands(hip, orc):
    """A large NumPy frame (radial map, order 1) is processed in bands of rows with uploads and downloads overlapped;
    the result must equal the one-shot path and the device-resident path, also when a band's source rows lie far away."""
    torch = pytest.importorskip("torch")
    old = hip.get_option("host_duplex")
    try:
        for k, (shape, a) in enumerate([((2100, 2048), (1000.0, 1100.0, list(configs.COEF_DOT_05))),
                                        ((2304, 1900), (-300.0, 2500.0, [1.0, 1e-4])),          # centre outside the frame
                                        ((2200, 2000), (900.0, 1000.0, [0.3, 9e-4])),           # strong: bands reach far rows
                                        ((2200, 2000), (900.0, 1000.0, [-1.0, 0.0]))]):         # point reflection: rows reversed
            img = noise(600 + k, shape)
            padded = np.zeros((shape[0], shape[1] + 37), np.float32)
            padded[:, :shape[1]] = img
            dev = {b: pp.unwarp_image_backward(torch.from_numpy(img).cuda(), *a, blend=b).cpu().numpy() for b in ("f64lerp", "scipy")}
            for mode in (2, 1, 0):       # forced banded path, probe-gated, one-shot
                hip.set_option("host_duplex", mode)
                assert np.array_equal(pp.unwarp_image_backward(img, *a, blend="f64lerp"), dev["f64lerp"]), (shape, mode)
                assert np.array_equal(pp.unwarp_image_backward(img, *a, blend="scipy"), dev["scipy"]), (shape, mode)
                assert np.array_equal(pp.unwarp_image_backward(padded[:, :shape[1]], *a, blend="f64lerp"), dev["f64lerp"]), (shape, mode)
            if k == 0:
                assert np.array_equal(dev["f64lerp"], orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, "f64lerp")))
        # perspective frames: the band's source rows come from its four corners
        img = noise(650, (2300, 2000))
        for coef in ([0.95, -0.02, 40.0, 0.03, 0.9, -25.0, 2e-5, -1e-5], [1.0, 0.0, 0.0, 0.0, -1.0, 2299.0, 0.0, 0.0],   # mild; rows flipped
                     [0.7, 0.4, -300.0, -0.4, 0.7, 900.0, 1e-4, 8e-5]):                                                    # rotated, strong
            devp = {o: pp.correct_perspective_image(torch.from_numpy(img).cuda(), coef, order=o).cpu().numpy() for o in (1, 0)}
            for mode in (2, 0):
                hip.set_option("host_duplex", mode)
                for o in (1, 0):
                    assert np.array_equal(pp.correct_perspective_image(img, coef, order=o, blend=DEV), devp[o]), (coef, mode, o)
            assert np.array_equal(devp[1], orc.correct_perspective_image(img, coef, blend=orc.BLEND_F64LERP))
            # the fused perspective -> radial map: the radial model over the band's rectangle of perspective positions
            for fact in ([1.0, 2e-5, -3e-9], [0.5, 6e-4]):
                devf = pp.unwarp_perspective_fused(torch.from_numpy(img).cuda(), 1020.0, 1130.0, fact, coef).cpu().numpy()
                for mode in (2, 0):
                    hip.set_option("host_duplex", mode)
                    assert np.array_equal(pp.unwarp_perspective_fused(img, 1020.0, 1130.0, fact, coef, blend=DEV), devf), (coef, fact, mode)
        # interleaved colour frames take the same banded route (util.unwarp_color_image_backward)
        from discorpy_amd.util import utility as util
        rgb = typed_image("uint8", (2400, 2400, 3), 640)
        a = (1150.0, 1260.0, [1.0, 3e-5, 2e-8])
        got = {}
        for mode in (2, 0):
            hip.set_option("host_duplex", mode)
            got[mode] = util.unwarp_color_image_backward(rgb, *a)
        assert np.array_equal(got[2], got[0])
        yd, xd = orc.radial_coords(2400, 2400, *a, poly=orc.POLY_KERNEL)
        assert np.array_equal(got[2][:, :, 1], orc.map_coordinates(np.ascontiguousarray(rgb[:, :, 1]), yd, xd, 1))
        # ... and so do large single-channel frames of the other element types
        u16 = typed_image("uint16", (3000, 3000), 641)
        for order in (1, 0):
            res = {}
            for mode in (2, 0):
                hip.set_option("host_duplex", mode)
                res[mode] = pp.unwarp_image_backward(u16, 1400.0, 1600.0, [1.0, 2e-5], order=order)
            assert res[2].dtype == np.uint16 and np.array_equal(res[2], res[0])
            assert np.array_equal(res[2], pp.unwarp_image_backward(torch.from_numpy(u16).cuda(), 1400.0, 1600.0, [1.0, 2e-5],
                                                                   order=order).cpu().numpy())
    finally:
        hip.set_option("host_duplex", old)


def test_out_argument_and_recycled_outputs(hip, orc):
    from discorpy_amd import _pool
    img = noise(71, (600, 700))                                   # 1.6 MiB: above the pool's threshold
    a = (333.0, 290.0, list(configs.COEF_DOT_05))
    want = orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, HOST))
    out = np.full(img.shape, -1.0, np.float32)
    assert pp.unwarp_image_backward(img, *a, out=out) is out and np.array_equal(out, want)
    _pool.clear()
    r1 = pp.unwarp_image_backward(img, *a)
    addr = r1.ctypes.data
    keep = r1.copy()
    del r1
    r2 = pp.unwarp_image_backward(img[::-1].copy(), *a)           # reuses the block r1 gave back
    assert r2.ctypes.data == addr and np.array_equal(keep, want) and not np.array_equal(r2, want)
    r3 = pp.unwarp_image_backward(img, *a)                        # r2 is alive: a different block
    assert r3.ctypes.data != addr and np.array_equal(r3, want)
    vol = noise(72, (4, 200, 300))
    sl = np.empty((4, 300), np.float32)
    assert np.array_equal(pp.unwarp_slice_backward(vol, 150.0, 100.0, [1.0, 1e-3], 77, out=sl),
                          orc.unwarp_slice_backward(vol, 150.0, 100.0, [1.0, 1e-3], 77, **kernel_oracle(orc, HOST)))
    ch = np.empty((4, 11, 300), np.float32)
    assert pp.unwarp_chunk_slices_backward(vol, 150.0, 100.0, [1.0, 1e-3], 50, 60, out=ch) is ch
    assert np.array_equal(ch, orc.unwarp_chunk_slices_backward(vol, 150.0, 100.0, [1.0, 1e-3], 50, 60,
                                                               **kernel_oracle(orc, HOST)))


def test_explicit_coordinates_match_oracle(hip, orc):
    img = noise(3, (70, 90))
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        ys = (rng.random(5000) * 75 - 3).astype(dt)            # includes out-of-image values: clamped
        xs = (rng.random(5000) * 96 - 3).astype(dt)
        for order, blend in [(0, "scipy"), (1, "scipy"), (1, "f64lerp"), (1, "f32")]:
            want = orc.remap_coords(img, ys, xs, order=order, blend=kernel_oracle(orc, blend)["blend"], mode="nearest")
            assert np.array_equal(pp.remap_coordinates(img, ys, xs, order=order, mode="nearest", blend=blend), want)
    assert pp.remap_coordinates(img, np.zeros((0,), np.float32), np.zeros((0,), np.float32)).shape == (0,)
    # coordinates outside the image follow scipy's `mode` at orders 0 and 1 -- compared with scipy itself, which is what the
    # reference hands map_index and mode to (postprocessing.py:489-491) -- for float32 and integer images, host and
    # float64 coordinates; the spline orders clamp them and say so
    import warnings
    from scipy.ndimage import map_coordinates
    ys = (rng.random(4000) * 70 * 7 - 70 * 3).astype(np.float32)
    xs = (rng.random(4000) * 90 * 7 - 90 * 3).astype(np.float32)
    ys[:60] = np.linspace(-2.0, 71.0, 60)
    xs[:60] = 44.25
    u16 = (img * 60000).astype(np.uint16)
    for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        for order in (0, 1):
            for yy, xx in ((ys, xs), (ys.astype(np.float64) * 1.0000001, xs.astype(np.float64))):
                with warnings.catch_warnings():
                    warnings.simplefilter("error")
                    got = pp.remap_coordinates(img, yy, xx, order=order, mode=mode, blend="scipy")
                    got16 = pp.remap_coordinates(u16, yy, xx, order=order, mode=mode)
                assert np.array_equal(got, map_coordinates(img, (yy, xx), order=order, mode=mode)), (mode, order, yy.dtype)
                assert np.array_equal(got, orc.remap_coords(img, yy, xx, order=order, mode=mode)), (mode, order)
                assert np.array_equal(got16, map_coordinates(u16, (yy, xx), order=order, mode=mode)), (mode, order, "uint16")
    # ... and at the spline orders (scipy's pre-padding for 'nearest' / 'grid-constant' included): equal to the oracle bit for
    # bit, which is equal to scipy (tests/test_oracle_golden.py)
    for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        for order in (2, 3, 4, 5):
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = pp.remap_coordinates(img, ys, xs, order=order, mode=mode)
                got16 = pp.remap_coordinates(u16, ys, xs, order=order, mode=mode)
            assert np.array_equal(got, orc.remap_coords(img, ys, xs, order=order, mode=mode)), (mode, order)
            ref = map_coordinates(img, (ys, xs), order=order, mode=mode)
            assert np.count_nonzero(got != ref) <= 2 and np.max(np.abs(got - ref)) <= 1e-6, (mode, order)
            assert np.array_equal(got16, orc.map_coordinates(u16, ys, xs, order, mode)), (mode, order, "uint16")


def test_explicit_coordinates_every_type_order_mode_on_degenerate_shapes(hip, orc):
    """remap_coordinates against the oracle (itself held to scipy in the CPU suite): eight element types, orders 0..5, eight
    modes, shapes down to 1 x 1, coordinates inside and up to three image sizes outside."""
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
                    got = pp.remap_coordinates(img, ys, xs, order=order, mode=mode, blend="scipy")
                    want = orc.map_coordinates(img, ys, xs, order, mode) if dt != np.float32 else \
                        orc.remap_coords(img, ys, xs, order=order, mode=mode)
                    assert got.dtype == want.dtype and np.array_equal(got, want), (shape, np.dtype(dt).name, mode, order)


def test_non_finite_pixels_next_to_a_clipped_edge_scipy_blend(hip, orc):
    """Coordinates clipped to the last column / row read (len - 1, len - 1 folded) in scipy: a NaN or Inf at len - 2 must not
    reach them through a zero weight.  The scipy blend reproduces that (staged and direct kernels, frames and stacks); compared
    with the oracle, which is scipy's arithmetic."""
    from scipy.ndimage import map_coordinates
    for shape in ((300, 517), (64, 4096)):
        h, w = shape
        img = noise(5, shape)
        img[:, w - 2] = np.nan
        img[h - 2, :] = np.inf
        img[7, 9] = -np.inf
        a = (img, 0.75 * w, 0.7 * h, [1.08, 2.0e-4])            # magnifying model: a band of pixels clips to the right / bottom edge
        want = orc.unwarp_image_backward(*a, poly=orc.POLY_KERNEL, blend=orc.BLEND_SCIPY)
        assert np.isfinite(want[:, -1]).sum() > h // 4              # (clipped pixels stay finite in scipy's arithmetic)
        for opts in ({}, {"x_wg_box": 0}, {"lds_gather": 0}):
            try:
                for k, v in opts.items():
                    hip.set_option(k, v)
                got = pp.unwarp_image_backward(*a, blend="scipy")
            finally:
                for k in opts:
                    hip.set_option(k, 1)
            assert np.array_equal(got, want, equal_nan=True), (shape, opts, hip.last_kernel())
        vol = np.stack([img, img[::-1].copy()])
        got = pp.unwarp_chunk_slices_backward(vol, a[1], a[2], a[3], h // 2, h - 1, blend="scipy")
        want = orc.unwarp_chunk_slices_backward(vol, a[1], a[2], a[3], h // 2, h - 1, poly=orc.POLY_KERNEL, blend=orc.BLEND_SCIPY)
        assert np.array_equal(got, want, equal_nan=True), shape


def test_slice_index_fractional_and_outside_for_every_element_type(hip, orc):
    """unwarp_slice_backward does not validate `index` (postprocessing.py:215): fractional and out-of-range values go through
    the same arithmetic.  Every element type against the oracle (held to the reference on 1000 such cases when it was fixed)."""
    rng = np.random.default_rng(55)
    for t in range(60):
        d, h, w = int(rng.integers(1, 3)), int(rng.integers(3, 70)), int(rng.integers(3, 70))
        dt = [np.float32, np.uint16, np.float64, np.uint8, np.int16][t % 5]
        vol = rng.random((d, h, w))
        vol = (vol * 60000).astype(dt) if dt == np.uint16 else (vol * 250).astype(dt) if dt == np.uint8 else \
            (vol * 60000 - 30000).astype(dt) if dt == np.int16 else vol.astype(dt)
        xc, yc = float(rng.uniform(-0.2 * w, 1.2 * w)), float(rng.uniform(-0.2 * h, 1.2 * h))
        fact = [1.0 + float(rng.uniform(-.1, .1)), float(rng.uniform(-3e-3, 3e-3)), float(rng.uniform(-1e-4, 1e-4))]
        idx = [float(rng.uniform(-5, h + 5)), int(rng.integers(-3, h + 3)), float(rng.integers(0, h)) + 0.5][t % 3]
        got = pp.unwarp_slice_backward(vol, xc, yc, fact, idx, blend="scipy")
        want = orc.unwarp_slice_backward(vol, xc, yc, fact, idx)
        assert got.dtype == np.float32 and np.array_equal(got, want), (t, np.dtype(dt).name, (d, h, w), idx)


def test_device_resident_tensors_take_the_same_path(hip, orc):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("torch sees no ROCm device on a GPU run")
    img = noise(2, (260, 410))
    a = (201.0, 133.0, list(configs.COEF_DOT_05))
    want = orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, "scipy"))
    t = torch.from_numpy(img).cuda()
    out = pp.unwarp_image_backward(t, *a, blend="scipy")
    assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == img.shape
    assert np.array_equal(out.cpu().numpy(), want)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                                  # kernels follow torch's current stream
        out2 = pp.unwarp_image_backward(t, *a, blend="scipy")
    s.synchronize()
    assert np.array_equal(out2.cpu().numpy(), want)
    chan = torch.from_numpy(noise(3, (64, 80, 3))).cuda()[:, :, 1]          # strided device view
    assert np.array_equal(pp.unwarp_image_backward(chan, 40.0, 30.0, [1.0, 1e-3], blend="scipy").cpu().numpy(),
                          orc.unwarp_image_backward(np.ascontiguousarray(chan.cpu().numpy()), 40.0, 30.0, [1.0, 1e-3],
                                                    **kernel_oracle(orc, "scipy")))
    vol = torch.from_numpy(noise(4, (5, 100, 140))).cuda()
    got = pp.unwarp_chunk_slices_backward(vol, 70.0, 50.0, [1.0, 2e-3], 20, 40, blend="scipy")
    assert got.is_cuda and np.array_equal(
        got.cpu().numpy(), orc.unwarp_chunk_slices_backward(vol.cpu().numpy(), 70.0, 50.0, [1.0, 2e-3], 20, 40,
                                                            **kernel_oracle(orc, "scipy")))


# --------------------------------------------------------------------------- element types other than float32

def typed_close(out, ref, order):
    """Bit-exact at every order for float32 and the integer types (the spline orders since round 2); float64 images differ
    in the last places of the double at the spline orders."""
    assert out.dtype == ref.dtype and out.shape == ref.shape, (out.dtype, ref.dtype, out.shape, ref.shape)
    if order <= 1 or out.dtype != np.float64:
        return np.array_equal(out, ref)
    return np.allclose(out, ref, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("dt", G12_DTYPES)
def test_g12_element_types_against_the_reference(hip, dt):
    """Output dtype = input dtype, scipy's integer rounding and saturation; the slice function returns float32."""
    g = golden("g12_dtypes40x52")
    im, vol = g12_inputs(g, dt)
    xc, yc, fact, coef = float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]), list(g["list_coef"])
    for order in (0, 1, 3):
        assert typed_close(pp.unwarp_image_backward(im, xc, yc, fact, order=order), g["radial_o%d_%s" % (order, dt)], order)
        assert typed_close(pp.remap_coordinates(im, g["pts_y"], g["pts_x"], order=order), g["points_o%d_%s" % (order, dt)], order)
    assert typed_close(pp.unwarp_image_backward(im, xc, yc, fact, order=2, mode="nearest"), g["radial_o2_nearest_" + dt], 2)
    assert typed_close(pp.correct_perspective_image(im, coef), g["persp_o1_" + dt], 1)
    assert typed_close(pp.correct_perspective_image(im, coef, order=5, mode="grid-wrap"), g["persp_o5_wrap_" + dt], 5)
    sl = pp.unwarp_slice_backward(vol, xc, yc, fact, int(g["index"]))
    assert sl.dtype == np.float32 and np.array_equal(sl, g["slice_" + dt])
    assert typed_close(pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, int(g["start"]), int(g["stop"])), g["chunk_" + dt], 1)


@pytest.mark.parametrize("dt", G12B_DTYPES)
def test_g12b_int64_uint64_bool_against_the_reference(hip, orc, dt):
    """Inputs the reference accepts because scipy does (postprocessing.py:147, 227, 251, 491): 64-bit integers -- read as doubles,
    stored as the reference's cast stores them on x86-64, extremes included -- and bool; orders 0 / 1 bit for bit against golden
    G12b, order 3 within the float64 noise of the recursive filter (test_oracle_golden.wide_close)."""
    from test_oracle_golden import wide_close
    g = golden("g12b_wide_types40x52")
    im = wide_image(dt, g["shape"], g["seed_" + dt])
    vol = wide_image(dt, g["vol_shape"], int(g["seed_" + dt]) + 100)
    xc, yc, fact, coef = float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]), list(g["list_coef"])
    for order in (0, 1, 3):
        imd = im.astype(np.float64)
        pre_r = orc.unwarp_image_backward(imd, xc, yc, fact, order=order) if order > 1 else None
        pre_p = orc.map_coordinates(imd, g["pts_y"], g["pts_x"], order) if order > 1 else None
        assert wide_close(pp.unwarp_image_backward(im, xc, yc, fact, order=order), g["radial_o%d_%s" % (order, dt)], order, pre_r), order
        assert wide_close(pp.remap_coordinates(im, g["pts_y"], g["pts_x"], order=order), g["points_o%d_%s" % (order, dt)], order, pre_p), order
    assert wide_close(pp.correct_perspective_image(im, coef), g["persp_o1_" + dt], 1)
    sl = pp.unwarp_slice_backward(vol, xc, yc, fact, int(g["index"]))
    assert sl.dtype == np.float32 and np.array_equal(sl, g["slice_" + dt])
    assert wide_close(pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, int(g["start"]), int(g["stop"])), g["chunk_" + dt], 1)
    # a larger frame against the oracle (tiles, bands of the host path), and the colour helper
    from discorpy_amd.util import utility as util
    big = wide_image(dt, (700, 900), 31)
    a = (430.5, 333.25, [1.0, -2e-5, 3e-8])
    yd, xd = orc.radial_coords(700, 900, *a, poly=orc.POLY_KERNEL)
    for order in (0, 1):
        assert np.array_equal(pp.unwarp_image_backward(big, *a, order=order), orc.map_coordinates(big, yd, xd, order))
    rgb = wide_image(dt, (120, 150, 3), 32)
    got = util.unwarp_color_image_backward(rgb, 70.0, 60.0, [1.0, 1e-4])
    yd, xd = orc.radial_coords(120, 150, 70.0, 60.0, [1.0, 1e-4], poly=orc.POLY_KERNEL)
    for c in range(3):
        assert np.array_equal(got[:, :, c], orc.map_coordinates(np.ascontiguousarray(rgb[:, :, c]), yd, xd, 1))


def test_complex_images_go_through_as_real_and_imaginary_parts(hip):
    g = golden("g12b_wide_types40x52")
    cim = (np.random.default_rng(861).random((40, 52)) + 1j * np.random.default_rng(862).random((40, 52))).astype(np.complex64)
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    out = pp.unwarp_image_backward(cim, *a, blend="scipy")
    assert out.dtype == np.complex64 and np.array_equal(out, g["radial_o1_complex64"])
    out = pp.correct_perspective_image(cim, list(g["list_coef"]), blend="scipy")
    assert out.dtype == np.complex64 and np.array_equal(out, g["persp_o1_complex64"])
    vol = np.stack([cim, cim[::-1]]).astype(np.complex128)
    ch = pp.unwarp_chunk_slices_backward(vol, *a, 3, 9)
    assert ch.dtype == np.complex128 and ch.shape == (2, 7, 52)
    assert np.array_equal(ch.real, pp.unwarp_chunk_slices_backward(np.ascontiguousarray(vol.real), *a, 3, 9))


@pytest.mark.parametrize("dt", ["uint8", "uint16", "int32", "float64", "float32"])
def test_element_types_match_oracle_on_ragged_shapes(hip, orc, dt):
    L = hip.lib()
    for k, (h, w) in enumerate([(1, 1), (1, 9), (7, 1), (63, 257), (300, 421)]):
        im = typed_image(dt, (h, w), 40 + k)
        a = (0.45 * w, 0.55 * h, [1.0, 2e-3 * 300 / max(h, w), 1e-6])
        for order in (0, 1):
            want = orc.map_coordinates(im, *orc.radial_coords(h, w, *a, poly=orc.POLY_KERNEL), order)
            if dt == "float32":     # the typed entry point also takes float32 (exact scipy blend)
                out = np.empty_like(im)
                fa, nf = hip.fact_array(a[2])
                hip.check(L.dcp_unwarp_image_typed(im.ctypes.data, out.ctypes.data, 0, h, w, w, 1, a[0], a[1], fa, nf,
                                                   order, 0, hip.MEM_HOST, -1, None))
            else:
                out = pp.unwarp_image_backward(im, *a, order=order)
            assert typed_close(out, want, order), (dt, h, w, order)
    # one channel of an interleaved colour image, and the whole image through the colour helper
    from discorpy_amd.util import utility as util
    if dt != "float32":
        rgb = typed_image(dt, (50, 70, 3), 77)
        a = (33.0, 26.0, [1.0, 3e-3])
        yd, xd = orc.radial_coords(50, 70, *a, poly=orc.POLY_KERNEL)
        got = util.unwarp_color_image_backward(rgb, *a)
        assert got.dtype == rgb.dtype and got.shape == rgb.shape
        for c in range(3):
            want = orc.map_coordinates(np.ascontiguousarray(rgb[:, :, c]), yd, xd, 1)
            assert np.array_equal(pp.unwarp_image_backward(rgb[:, :, c], *a), want)
            assert np.array_equal(got[:, :, c], want)
        f = orc.map_coordinates(rgb[:, :, 0].copy(), *orc.radial_coords(50, 70, *a, poly=orc.POLY_KERNEL), 1)
        coef = [0.97, 0.01, 1.0, -0.02, 0.98, 0.5, 1e-4, -2e-4]
        py, px = pp.generate_fused_map((50, 70), *a, coef)
        fused = pp.unwarp_perspective_fused(rgb[:, :, 0], *a, coef)
        assert fused.dtype == rgb.dtype and np.array_equal(fused, orc.map_coordinates(rgb[:, :, 0].copy(), py, px, 1))


def test_integration_stub_of_the_docs_runs(hip, orc, monkeypatch):
    """The ctypes stub INTEGRATION.md shows a discorpy maintainer is executed as written."""
    import os
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def unwarp_image(" in b and "dcp_unwarp_image_f32.argtypes" in b)
    monkeypatch.setenv("DISCORPY_HIP_LIB", hip.LIB_PATH)
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    img = noise(90, (120, 160))
    a = (75.0, 61.0, [1.0, 2e-3, 1e-6])
    assert ns["available"](img, 1)
    got = ns["unwarp_image"](img, *a, 1)
    assert np.array_equal(got, orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, "scipy")))        # (the stub passes DCP_BLEND_SCIPY)
    with pytest.raises(ValueError):
        ns["_check"](ns["lib"]().dcp_unwarp_image_f32(None, None, 4, 4, 4, 1, 0.0, 0.0, None, 0, 1, 1, 1, 0, -1, None))
    # the continuation of the stub: many images, every image its own calibration, in one call
    more = next(b for b in blocks if "def unwarp_images(" in b)
    exec(compile(more, "INTEGRATION.md", "exec"), ns)
    frames = [noise(91 + i, (120, 160)) for i in range(3)]
    cals = [(75.0, 61.0, [1.0, 2e-3, 1e-6]), (80.5, 58.0, [0.99, 1e-3]), (70.0, 66.25, [1.01, -1e-3, 2e-6, 1e-9])]
    outs = ns["unwarp_images"](frames, [c[0] for c in cals], [c[1] for c in cals], [c[2] for c in cals])
    for f, c, o in zip(frames, cals, outs):
        assert np.array_equal(o, orc.unwarp_image_backward(f, *c, **kernel_oracle(orc, "scipy")))


def test_release_scratch_then_work_again(hip, orc):
    img = noise(95, (300, 400))
    a = (190.0, 140.0, [1.0, 1e-3])
    before = pp.unwarp_image_backward(img, *a, order=3)
    pp.unwarp_chunk_slices_backward(noise(96, (3, 100, 120)), 60.0, 50.0, [1.0, 1e-3], 10, 30)
    hip.release_scratch()
    hip.release_scratch()                                  # idempotent
    assert np.array_equal(pp.unwarp_image_backward(img, *a, order=3), before)
    assert np.array_equal(pp.unwarp_image_backward(img, *a), orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, HOST)))


def test_c_abi_from_plain_c(hip, orc, tmp_path):
    """tests/c/abi_smoke.c: the boundary used as a C library (gcc, no Python in the loop, no HIP headers) -- host and
    device pointers, a uint16 stack, error codes -- checked against the oracle inside the C program."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "abi_smoke")
    libdir, orcdir = os.path.dirname(hip.LIB_PATH), os.path.join(ROOT, "oracle")
    cmd = ["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", orcdir, os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-o", exe, "-L", libdir, "-ldiscorpy_hip", "-L", orcdir, "-lunwarp_oracle", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + orcdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    res = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0 and "abi_smoke ok" in res.stdout, (res.returncode, res.stdout, res.stderr)


class CudaArrayInterfaceOnly:
    """What a CuPy / Numba device array looks like to this package: shape, dtype and __cuda_array_interface__."""

    def __init__(self, tensor):
        self._t = tensor
        self.__cuda_array_interface__ = tensor.__cuda_array_interface__
        self.shape = tuple(tensor.shape)
        self.dtype = np.dtype(self.__cuda_array_interface__["typestr"])


def test_cuda_array_interface_inputs(hip, orc):
    torch = pytest.importorskip("torch")
    img = noise(81, (150, 210))
    a = (101.0, 77.0, [1.0, 1.5e-3])
    want = orc.unwarp_image_backward(img, *a, **kernel_oracle(orc, "f64lerp"))
    t = torch.from_numpy(img).cuda()
    res = pp.unwarp_image_backward(CudaArrayInterfaceOnly(t), *a)
    assert isinstance(res, hip.DeviceArray) and res.shape == img.shape and res.dtype == np.float32
    torch.cuda.synchronize()
    assert np.array_equal(res.copy_to_host(), want)
    assert np.array_equal(torch.as_tensor(res, device="cuda").cpu().numpy(), want)          # wrapped without a copy
    strided = CudaArrayInterfaceOnly(torch.from_numpy(noise(82, (150, 210, 2))).cuda()[:, :, 1])
    got = pp.unwarp_image_backward(strided, *a)
    assert np.array_equal(got.copy_to_host(), orc.unwarp_image_backward(np.ascontiguousarray(strided._t.cpu().numpy()), *a,
                                                                         **kernel_oracle(orc, "f64lerp")))
    out = torch.empty((150, 210), dtype=torch.float32, device="cuda")
    assert pp.unwarp_image_backward(CudaArrayInterfaceOnly(t), *a, out=CudaArrayInterfaceOnly(out)) is not None
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    vol = typed_image("uint16", (4, 90, 120), 83)
    v = CudaArrayInterfaceOnly(torch.from_numpy(vol).cuda())
    ch = pp.unwarp_chunk_slices_backward(v, 60.0, 45.0, [1.0, 1e-3], 10, 40)
    assert ch.dtype == np.uint16 and np.array_equal(ch.copy_to_host(), orc.unwarp_chunk_slices_backward(vol, 60.0, 45.0, [1.0, 1e-3], 10, 40,
                                                                                                        poly=orc.POLY_KERNEL))
    sl = pp.unwarp_slice_backward(v, 60.0, 45.0, [1.0, 1e-3], 33)
    assert sl.shape == (4, 120) and sl.dtype == np.float32
    assert np.array_equal(sl.copy_to_host(), orc.unwarp_slice_backward(vol, 60.0, 45.0, [1.0, 1e-3], 33, poly=orc.POLY_KERNEL))


def test_cuda_array_interface_image_with_map_index_and_explicit_coordinates(hip, orc):
    """a CuPy / Numba image with map_index= (host maps, device maps, the maps _generate_perspective_map returns for
    such an image) and remap_coordinates with host or device coordinates -- the kernel must never be handed a host pointer
    for a device image."""
    torch = pytest.importorskip("torch")
    g = golden("g3_perspective64")
    coef = list(g["coef_backward"])
    t = torch.from_numpy(g["mat"]).cuda()
    dev_img = CudaArrayInterfaceOnly(t)
    ymap, xmap = pp._generate_perspective_map(dev_img, coef)
    assert isinstance(ymap, hip.DeviceArray) and ymap.shape == (64 * 64, 1)        # device-resident maps for a device image
    yh, xh = pp._generate_perspective_map(g["mat"], coef)
    assert np.array_equal(ymap.copy_to_host(), yh) and np.array_equal(xmap.copy_to_host(), xh)
    for maps in ((ymap, xmap), (yh, xh), (CudaArrayInterfaceOnly(torch.from_numpy(yh).cuda()), CudaArrayInterfaceOnly(torch.from_numpy(xh).cuda()))):
        res = pp.correct_perspective_image(dev_img, coef, map_index=maps, blend="scipy")
        torch.cuda.synchronize()
        assert np.array_equal(res.copy_to_host(), g["cor_backward"])
    g9 = golden("g9_points33x47")
    im = noise(g9["seed"], g9["shape"])
    d9 = CudaArrayInterfaceOnly(torch.from_numpy(im).cuda())
    got = pp.remap_coordinates(d9, g9["ys"], g9["xs"], blend="scipy")                          # host float32 coordinates
    assert np.array_equal(got.copy_to_host(), g9["out_order1"])
    got = pp.remap_coordinates(d9, g9["ys64"], g9["xs64"], blend="scipy")                      # host float64 coordinates
    assert np.array_equal(got.copy_to_host(), g9["out64_order1"])
    ys_d = CudaArrayInterfaceOnly(torch.from_numpy(g9["ys64"]).cuda())
    xs_d = CudaArrayInterfaceOnly(torch.from_numpy(g9["xs64"]).cuda())
    assert np.array_equal(pp.remap_coordinates(d9, ys_d, xs_d, order=0).copy_to_host(), g9["out64_order0"])   # device coordinates
    assert np.array_equal(pp.remap_coordinates(d9, ys_d, g9["xs64"], order=0).copy_to_host(), g9["out64_order0"])   # mixed


def test_device_views_with_16_megabyte_row_strides(hip, orc):
    """the tuned kernels form row offsets with 24-bit multiplies; a device view whose rows are 2^22 elements apart
    -- vol[:, k, :] of a (depth, 4096, 1024) volume -- must take the kernels with full 32-bit products (frames and stacks)."""
    torch = pytest.importorskip("torch")
    vol = torch.from_numpy(noise(91, (24, 4096, 1024))).cuda()
    a = (500.5, 11.25, [1.0, -1.0e-4, 2.0e-7])
    for k in (0, 1777, 4095):
        view = vol[:, k, :]
        assert view.stride(0) == 1 << 22 and view.shape == (24, 1024)
        host = np.ascontiguousarray(view.cpu().numpy())
        for order in (0, 1):
            got = pp.unwarp_image_backward(view, *a, order=order).cpu().numpy()
            assert "strided" in hip.last_kernel()
            assert np.array_equal(got, orc.unwarp_image_backward(host, *a, order=order, **kernel_oracle(orc, "f64lerp")))
    # a stack whose projections' rows are 2^22 elements apart: (depth, rows, width) = vol[:, ::?] is not expressible with a
    # row stride that large and 24 projections in 4 GiB, so one projection with huge rows: a (1, 24, 1024) stack view
    st = vol[:, 100, :].unsqueeze(0)
    assert st.stride(1) == 1 << 22
    got = pp.unwarp_chunk_slices_backward(st, *a, 2, 20).cpu().numpy()
    assert np.array_equal(got, orc.unwarp_chunk_slices_backward(np.ascontiguousarray(st.cpu().numpy()), *a, 2, 20, **kernel_oracle(orc, "f64lerp")))


def test_element_types_on_device_tensors_and_stacks(hip, orc):
    torch = pytest.importorskip("torch")
    a = (70.0, 50.0, [1.0, 2e-3])
    for dt in ("uint8", "int16", "float64"):
        vol = typed_image(dt, (6, 100, 140), 5)
        tv = torch.from_numpy(vol).cuda()
        got = pp.unwarp_chunk_slices_backward(tv, *a, 20, 40)
        assert got.is_cuda and str(got.dtype) == "torch." + dt
        assert np.array_equal(got.cpu().numpy(), orc.unwarp_chunk_slices_backward(vol, *a, 20, 40, poly=orc.POLY_KERNEL))
        sl = pp.unwarp_slice_backward(tv, *a, 33)
        assert sl.dtype == torch.float32
        assert np.array_equal(sl.cpu().numpy(), orc.unwarp_slice_backward(vol, *a, 33, poly=orc.POLY_KERNEL))
        im = torch.from_numpy(vol[0]).cuda()
        for order in (1, 3):
            out = pp.unwarp_image_backward(im, *a, order=order, mode="mirror")
            want = orc.map_coordinates(vol[0], *orc.radial_coords(100, 140, *a, poly=orc.POLY_KERNEL), order, "mirror")
            assert str(out.dtype) == "torch." + dt and typed_close(out.cpu().numpy(), want, order)
    # a host stack with padded rows / projections (the row-band staging of the typed path)
    big = typed_image("uint16", (4, 90, 130), 6)
    view = big[:, 5:85, 7:120]
    assert np.array_equal(pp.unwarp_chunk_slices_backward(view, 50.0, 40.0, [1.0, 1e-3], 10, 70),
                          orc.unwarp_chunk_slices_backward(np.ascontiguousarray(view), 50.0, 40.0, [1.0, 1e-3], 10, 70,
                                                           poly=orc.POLY_KERNEL))


def test_calls_from_several_python_threads(hip, orc):
    """SURVEY 8(b) threading: the native layer is re-entrant -- per-thread staging, thread-local error text."""
    import threading
    imgs = [noise(200 + k, (300 + 7 * k, 400 - 5 * k)) for k in range(6)]
    a = [(150.0 + k, 120.0 - k, [1.0, 1e-3 * (k + 1) / 6, 2e-6]) for k in range(6)]
    want = [orc.unwarp_image_backward(im, *p, **kernel_oracle(orc, HOST)) for im, p in zip(imgs, a)]
    want3 = [pp.unwarp_image_backward(im, *p, order=3, mode="mirror") for im, p in zip(imgs, a)]   # one shared spline workspace
    vol = noise(300, (5, 120, 160))
    want_c = orc.unwarp_chunk_slices_backward(vol, 80.0, 60.0, [1.0, 2e-3], 30, 60, **kernel_oracle(orc, HOST))
    errors = []

    def work(k):
        try:
            for _ in range(8):
                assert np.array_equal(pp.unwarp_image_backward(imgs[k], *a[k]), want[k])
                assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, 80.0, 60.0, [1.0, 2e-3], 30, 60), want_c)
                assert np.array_equal(pp.unwarp_image_backward(imgs[k], *a[k], order=3, mode="mirror"), want3[k])
                with pytest.raises(ValueError, match="nfact"):
                    hip.check(hip.lib().dcp_unwarp_image_f32(imgs[k].ctypes.data, imgs[k].ctypes.data, 4, 4, 4, 1, 0.0, 0.0,
                                                             hip.fact_array([1.0])[0], 99, 1, 1, 1, hip.MEM_HOST, -1, None))
        except Exception as e:            # noqa: BLE001 -- reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_frame_larger_than_4_gib(hip, orc):
    """A 33 000 x 33 000 float32 frame (4.36 GB) is beyond the 32-bit offsets of the tuned kernels; the C ABI sends it
    to the generic kernels (64-bit addressing, exact blend) instead of refusing it.  Rows at the top, the middle and
    the bottom are checked against the oracle."""
    H = W = 33000
    base = noise(5, (500, W))
    img = np.tile(base, (H // 500, 1))
    assert img.shape == (H, W) and img.nbytes > 2 ** 32
    a = (W * 0.47, H * 0.52, [1.0, 2.0e-6, -1.5e-10, 2.0e-15])
    out = pp.unwarp_image_backward(img, *a)
    out0 = pp.unwarp_image_backward(img, *a, order=0)
    for r0 in (0, 16490, H - 24):
        want = orc.unwarp_stack_rows(img[None], *a, r0, 24, coord_round_f32=True, poly=orc.POLY_KERNEL, blend=orc.BLEND_SCIPY)[0]
        assert np.array_equal(out[r0:r0 + 24], want), r0
    del out
    # order 0 copies source pixels: every value of an output row is a value of the source rows its band reaches
    assert out0.shape == (H, W)
    for r0 in (3, 20000, H - 2):
        b0, bn = hip.stack_row_band(H, W, *a, r0, 1)
        assert np.isin(out0[r0], img[b0:b0 + bn]).all(), r0


def test_randomised_differential_campaign(hip, orc):
    """600 seeded random cases of tools/fuzz_parity.py (shapes, centres, models from mild to folding, homographies,
    strides, blends, stacks, coordinates, element types, spline orders and modes): HIP == oracle, bit for bit at
    orders 0/1.  profiles/rounds_1-4/r01c_fuzz_parity.txt holds a 32 000-case run."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(777)
    kinds = set()
    for k in range(600):
        kinds.add(fz.one_case(rng, k))
    assert kinds == {"radial", "persp", "fused", "stack", "coords", "spline", "color", "batch", "centres"}


# --------------------------------------------------------------------------- (c) BASELINE sizes

def test_cfg1_full_frame_equals_the_reference(hip, orc):
    """BASELINE config 1 at full size: unwarp_image_backward on the decoded data/dot_pattern_05.jpg (shipped as uint8 --
    JPEG is never decoded here) with data/coef_dot_05.txt, the call of examples/example_02.py:81.  SHA-256 of the whole
    float32 output == the reference's, for the scipy blend, the default blend and order 0; G4's rows and statistics."""
    import hashlib
    g, g4 = golden("g4b_dot_pattern_05_full"), golden("g4_dot_pattern_05")
    img = g["frame_u8"].astype(np.float32)
    a = (img, float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    assert a[1:] == (configs.XCENTER_DOT_05, configs.YCENTER_DOT_05, list(configs.COEF_DOT_05))
    for blend in ("scipy", "f64lerp"):
        out = pp.unwarp_image_backward(*a, blend=blend)
        assert out.shape == (800, 1280) and out.dtype == np.float32
        assert np.array_equal(out[g4["full_rows"]], g4["full_out_rows"]) and np.array_equal(out[::32, ::32], g["out_lattice"])
        assert out[400, 640] == np.float32(213.037353515625) and out[10, 10] == np.float32(195.14744567871094)
        assert out[799, 1279] == np.float32(89.43257141113281)
        assert abs(float(out.mean(dtype=np.float64)) - float(g4["full_stats"][3])) < 1e-4
        assert hashlib.sha256(out.tobytes()).digest() == g["out_sha256"].tobytes(), blend
    out0 = pp.unwarp_image_backward(*a, order=0)
    assert hashlib.sha256(out0.tobytes()).digest() == g["out_order0_sha256"].tobytes()
    # the uint8 frame itself (output dtype = input dtype, scipy's rounding): against the oracle
    u8 = pp.unwarp_image_backward(g["frame_u8"], *a[1:])
    assert u8.dtype == np.uint8 and np.array_equal(u8, orc.unwarp_image_backward(g["frame_u8"], *a[1:], poly=orc.POLY_KERNEL))


def test_cfg2_full_frame_against_oracle_and_properties(hip, orc):
    c = configs.cfg2()
    h, w = c["shape"]
    img = noise(c["seed"], (h, w))
    a = (img, c["xcenter"], c["ycenter"], c["list_fact"])
    out = pp.unwarp_image_backward(*a)
    want = orc.unwarp_image_backward(*a, **kernel_oracle(orc, HOST))
    assert np.array_equal(out, want)
    assert np.array_equal(pp.unwarp_image_backward(*a, order=0),
                          orc.unwarp_image_backward(*a, order=0, poly=orc.POLY_KERNEL))
    # reference arithmetic order (numpy polynomial, scipy blend): same pixels up to rounding-boundary
    # coordinates, of which SURVEY.md section 7 expects ~0-2 per frame
    ref = orc.unwarp_image_backward(*a, poly=orc.POLY_NUMPY, blend=orc.BLEND_SCIPY)
    differing = int(np.count_nonzero(ulp_diff(out, ref) > 1))
    assert differing <= 4, differing
    # identity model: every coordinate is an integer, the frame must come back bit for bit
    assert np.array_equal(pp.unwarp_image_backward(img, 17.0, 4000.5, [1.0]), img)
    # a constant frame stays constant whatever the map
    const = np.full((h, w), np.float32(0.3))
    assert np.array_equal(pp.unwarp_image_backward(const, c["xcenter"], c["ycenter"], c["list_fact"]), const)
    # chunk rows of a 1-projection stack == the same rows of the image result
    rows = pp.unwarp_chunk_slices_backward(img[None], c["xcenter"], c["ycenter"], c["list_fact"], 1000, 1015)
    assert np.array_equal(rows[0], out[1000:1016])


def test_cfg3_fused_full_frame(hip, orc):
    c = configs.cfg3()
    h, w = c["shape"]
    img = noise(c["seed"] + 1, (h, w))
    out = pp.unwarp_perspective_fused(img, c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"])
    want = orc.unwarp_fused(img, c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"],
                            **kernel_oracle(orc, HOST))
    assert np.array_equal(out, want)
    # integer translation through the homography: out[y, x] = in[y + 7, x + 5], edges replicated
    shift = [1.0, 0.0, 5.0, 0.0, 1.0, 7.0, 0.0, 0.0]
    moved = pp.correct_perspective_image(img, shift)
    assert np.array_equal(moved[:h - 7, :w - 5], img[7:, 5:])
    assert np.array_equal(moved[h - 7:, :w - 5], np.broadcast_to(img[h - 1, 5:], (7, w - 5)))


def test_cfg5_nine_term_8192_frame(hip, orc):
    c = configs.cfg5()
    h, w = c["shape"]
    img = noise(c["seed"], (h, w))
    a = (img, c["xcenter"], c["ycenter"], c["list_fact"])
    out = pp.unwarp_image_backward(*a)
    assert np.array_equal(out, orc.unwarp_image_backward(*a, **kernel_oracle(orc, HOST)))
    hip.set_option("x_coef_lds", 1)                               # LDS-staged coefficients: same bits
    try:
        assert np.array_equal(pp.unwarp_image_backward(*a), out)
    finally:
        hip.set_option("x_coef_lds", 0)


def test_cfg4_stack_sample(hip, orc):
    """A (12, 2560, 2560) sample of config 4: one sinogram and a 16-row chunk."""
    c = configs.cfg4(depth=12)
    vol = noise(c["seed"], c["shape"])
    a = (c["xcenter"], c["ycenter"], c["list_fact"])
    assert np.array_equal(pp.unwarp_slice_backward(vol, *a, 1277),
                          orc.unwarp_slice_backward(vol, *a, 1277, **kernel_oracle(orc, HOST)))
    assert np.array_equal(pp.unwarp_chunk_slices_backward(vol, *a, 2000, 2015),
                          orc.unwarp_chunk_slices_backward(vol, *a, 2000, 2015, **kernel_oracle(orc, HOST)))


# --------------------------------------------------------------------------- (d) the benched calls themselves, at full size

def _device_call(hip, fn, img, *args):
    """fn(src_ptr, dst_ptr) on device-resident buffers -- bench.py's call form (mem_kind = DEVICE, default stream)."""
    H, W = img.shape
    src = hip.DeviceBuffer(img.nbytes).upload(img)
    dst = hip.DeviceBuffer(img.nbytes)
    hip.check(fn(src.ptr, dst.ptr))
    out = dst.download((H, W), np.float32)
    src.free()
    dst.free()
    return out


@pytest.mark.parametrize("blend", ["f64lerp", "scipy"])
def test_cfg2_device_resident_call_is_the_benched_kernel_and_equals_the_oracle(hip, orc, blend):
    """BASELINE config 2 exactly as bench.py times it: dcp_unwarp_image_f32 on device pointers, 4096 x 4096, default options
    -> remap_wg_kernel; every pixel against the oracle.  Then the NumPy boundary with the banded host path forced off and on."""
    c = configs.cfg2()
    H, W = c["shape"]
    img = noise(c["seed"] + 7, (H, W))
    fa, nf = hip.fact_array(c["list_fact"])
    L = hip.lib()
    b = hip.BLEND_BY_NAME[blend]
    out = _device_call(hip, lambda s, d: L.dcp_unwarp_image_f32(s, d, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, 1, 1, b,
                                                                 hip.MEM_DEVICE, -1, None), img)
    assert hip.last_kernel() == "remap_wg_kernel<Radial,NF=5,%s>" % blend
    want = orc.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], **kernel_oracle(orc, blend))
    assert np.array_equal(out, want)
    old = hip.get_option("host_duplex")
    try:
        for mode in (0, 2):                         # one-shot upload / kernel / download, and the banded full-duplex path
            hip.set_option("host_duplex", mode)
            assert np.array_equal(pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], blend=blend), want), mode
    finally:
        hip.set_option("host_duplex", old)
    # the two other staged kernels on the same frame: one box per wave tile with the certificate, and with the per-pixel vote
    for opt, name in (("x_wg_box", "remap_lds_kernel<Radial,NF=5,%s,certified>" % blend), ("tile_cert", "remap_lds_kernel<Radial,NF=-1,%s,vote>" % blend)):
        hip.set_option(opt, 0)
        try:
            got = _device_call(hip, lambda s, d: L.dcp_unwarp_image_f32(s, d, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, 1, 1, b,
                                                                         hip.MEM_DEVICE, -1, None), img)
            assert hip.last_kernel() == name and np.array_equal(got, want), name
        finally:
            hip.set_option(opt, 1)


def test_cfg3_device_resident_calls_equal_the_oracle(hip, orc):
    """BASELINE config 3 as bench.py times it: the fused map, the perspective map alone and the reference's two passes."""
    c = configs.cfg3()
    H, W = c["shape"]
    img = noise(c["seed"] + 8, (H, W))
    fa, nf = hip.fact_array(c["list_fact"])
    ca, _ = hip.fact_array(c["list_coef"])
    L = hip.lib()
    fused = _device_call(hip, lambda s, d: L.dcp_unwarp_fused_f32(s, d, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, ca, 1,
                                                                   hip.BLEND_F64LERP, hip.MEM_DEVICE, -1, None), img)
    # (round 5) a tame homography in front of a radial model of certified curvature: the workgroup-box kernel, no per-pixel vote --
    # also where the inner clip cuts through a tile (config 3's homography maps the frame's right and bottom edges outside it)
    assert hip.last_kernel() == "remap_wg_kernel<Fused,NF=5,f64lerp>", hip.last_kernel()
    want_fused = orc.unwarp_fused(img, c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"], **kernel_oracle(orc, "f64lerp"))
    assert np.array_equal(fused, want_fused)
    # the third link of the parity chain at full size: HIP == oracle(kernel order) above, oracle(numpy order,
    # scipy blend) == reference on the golden sets (test_oracle_golden.py) -- and here kernel order against numpy order on all
    # 16.8 M pixels of config 3, whose map carries a second float32 round trip: pixels further than one float32 ulp apart are the
    # ones whose coordinate sits on a float32 rounding boundary
    ref_order = orc.unwarp_fused(img, c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"], poly=orc.POLY_NUMPY, blend=orc.BLEND_SCIPY)
    differing = int(np.count_nonzero(ulp_diff(fused, ref_order) > 1))
    print("cfg3 fused 4096^2: %d pixels further than 1 ulp from the reference's operation order" % differing)
    assert differing <= 8, differing
    hip.set_option("x_fused_wg", 0)               # rounds 1-4: one box per wave tile, every pixel voting on it -- the same pixels
    try:
        voted = _device_call(hip, lambda s, d: L.dcp_unwarp_fused_f32(s, d, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, ca, 1,
                                                                       hip.BLEND_F64LERP, hip.MEM_DEVICE, -1, None), img)
        assert hip.last_kernel().startswith("remap_lds_kernel<Fused,NF=5,f64lerp") and np.array_equal(voted, want_fused)
    finally:
        hip.set_option("x_fused_wg", 1)
    persp = _device_call(hip, lambda s, d: L.dcp_perspective_image_f32(s, d, H, W, W, 1, ca, 1, hip.BLEND_F64LERP, hip.MEM_DEVICE, -1, None), img)
    assert hip.last_kernel() == "remap_wg_kernel<Persp,NF=-1,f64lerp>"
    assert np.array_equal(persp, orc.correct_perspective_image(img, c["list_coef"], blend=orc.BLEND_F64LERP))
    old = hip.get_option("host_duplex")
    try:
        for mode in (0, 2):
            hip.set_option("host_duplex", mode)
            assert np.array_equal(pp.unwarp_perspective_fused(img, c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"], blend=DEV), fused), mode
            assert np.array_equal(pp.correct_perspective_image(img, c["list_coef"], blend=DEV), persp), mode
    finally:
        hip.set_option("host_duplex", old)


def test_cfg5_device_resident_call_equals_the_oracle(hip, orc):
    c = configs.cfg5()
    H, W = c["shape"]
    img = noise(c["seed"] + 9, (H, W))
    fa, nf = hip.fact_array(c["list_fact"])
    L = hip.lib()
    out = _device_call(hip, lambda s, d: L.dcp_unwarp_image_f32(s, d, H, W, W, 1, c["xcenter"], c["ycenter"], fa, nf, 1, 1, hip.BLEND_F64LERP,
                                                                 hip.MEM_DEVICE, -1, None), img)
    assert "NF=9" in hip.last_kernel() and "vote" not in hip.last_kernel()          # a certified, inline-coefficient kernel
    want = orc.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], **kernel_oracle(orc, "f64lerp"))
    assert np.array_equal(out, want)
    # the third link at full size: the kernel's even / odd Horner against numpy's sum of a_i * pow(ru, i) with
    # i up to 8 -- where the two orders are furthest apart -- on all 67 M pixels (134 M coordinates) of config 5
    ref_order = orc.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], poly=orc.POLY_NUMPY, blend=orc.BLEND_SCIPY)
    differing = int(np.count_nonzero(ulp_diff(out, ref_order) > 1))
    oy, ox = orc.radial_coords(H, W, c["xcenter"], c["ycenter"], c["list_fact"], poly=orc.POLY_KERNEL)
    ny, nx = orc.radial_coords(H, W, c["xcenter"], c["ycenter"], c["list_fact"], poly=orc.POLY_NUMPY)
    flips = int((oy.astype(np.float32) != ny.astype(np.float32)).sum() + (ox.astype(np.float32) != nx.astype(np.float32)).sum())
    print("cfg5 8192^2, 9 terms: %d pixels further than 1 ulp from the reference's operation order; %d of 134 M float32 coordinates differ" % (differing, flips))
    assert differing <= 16 and flips <= 16, (differing, flips)
    del ref_order, oy, ox, ny, nx
    old = hip.get_option("host_duplex")
    try:
        for mode in (0, 2):
            hip.set_option("host_duplex", mode)
            assert np.array_equal(pp.unwarp_image_backward(img, c["xcenter"], c["ycenter"], c["list_fact"], blend=DEV), want), mode
    finally:
        hip.set_option("host_duplex", old)


def test_cfg4_shard_all_rows_takes_the_staged_stack_kernel(hip, orc):
    """One 8-GPU shard's worth of config 4's geometry -- (64, 2560, 2560), ALL 2560 rows, default options, device-resident:
    the launcher must pick the workgroup-box stack kernel by itself (the kernel behind the whole-stack number), and every
    voxel must equal the oracle; the same through round 1's per-wave-box kernel (stack_wg = 0), which it must select then."""
    c = configs.cfg4(64)
    D, H, W = c["shape"]
    vol = noise(c["seed"] + 3, (D, H, W))
    fa, nf = hip.fact_array(c["list_fact"])
    L = hip.lib()
    src = hip.DeviceBuffer(vol.nbytes).upload(vol)
    dst = hip.DeviceBuffer(D * H * W * 4)
    want = orc.unwarp_stack_rows(vol, c["xcenter"], c["ycenter"], c["list_fact"], 0, H, coord_round_f32=True,
                                 **kernel_oracle(orc, "f64lerp"))
    for stack_wg, name in ((1, "stack_wg_kernel<NF=5,f64lerp>"), (0, "stack_lds_kernel<NF=5,f64lerp>")):
        hip.set_option("x_stack_wg", stack_wg)
        try:
            hip.debug_counters()
            hip.check(L.dcp_memcpy(dst.ptr, src.ptr, 4096, hip.COPY_D2D, -1, None))        # (scribble: the result must be rewritten)
            hip.check(L.dcp_unwarp_stack_rows_f32(src.ptr, dst.ptr, D, H, W, H * W, W, c["xcenter"], c["ycenter"], fa, nf, 0.0, H, 1,
                                                  hip.BLEND_F64LERP, hip.MEM_DEVICE, -1, None))
            assert hip.last_kernel() == name
            nofit, vote = hip.debug_counters()
            assert nofit == 0 and vote <= 64             # staged throughout (a handful of tiles may fail the zero-margin vote of the per-wave kernel)
            assert np.array_equal(dst.download((D, H, W), np.float32), want), name
        finally:
            hip.set_option("x_stack_wg", 1)
    src.free()
    dst.free()


def test_stack_wg_kernel_shapes_blends_and_integer_types(hip, orc):
    """stack_wg_kernel beyond the benched geometry: ragged tiles and depth chunks, rows not starting at 0, every blend, a
    9-term model (coefficients from LDS), strided projections, and the 8- / 16- / 32-bit integer and float64 instantiations (stack_wg = 2: also for
    launches this small)."""
    torch = pytest.importorskip("torch")
    D, H, W = 21, 300, 517
    vol = noise(501, (D, H, W))
    a = (250.3, 140.8, [1.0, 3.0e-5, -4.0e-8])
    hip.set_option("x_stack_wg", 2)                         # also for launches this small
    try:
        for blend in ("f64lerp", "scipy", "f32"):
            got = pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), *a, 7, 291, blend=blend).cpu().numpy()
            # (three terms: the NF = 4 instantiation on a zero-padded vector -- fma(r2, 0, a) = a exactly, so still bit-equal)
            assert hip.last_kernel().startswith("stack_wg_kernel<NF=4"), hip.last_kernel()
            assert np.array_equal(got, orc.unwarp_chunk_slices_backward(vol, *a, 7, 291, **kernel_oracle(orc, blend))), blend
        nine = (250.3, 140.8, [1.0, 1e-5, -2e-8, 1e-11, -3e-14, 2e-17, 1e-20, -1e-23, 1e-26])
        got = pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), *nine, 0, 299).cpu().numpy()
        assert hip.last_kernel().startswith("stack_wg_kernel<NF=-1"), hip.last_kernel()
        assert np.array_equal(got, orc.unwarp_chunk_slices_backward(vol, *nine, 0, 299, **kernel_oracle(orc, "f64lerp")))
        five = (250.3, 140.8, [1.0, 3.0e-5, -4.0e-8, 1e-11, -2e-14])
        got = pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), *five, 0, 299).cpu().numpy()
        assert hip.last_kernel() == "stack_wg_kernel<NF=5,f64lerp>"
        assert np.array_equal(got, orc.unwarp_chunk_slices_backward(vol, *five, 0, 299, **kernel_oracle(orc, "f64lerp")))
        big = torch.from_numpy(noise(502, (D, H + 9, W + 11))).cuda()
        view = big[:, 4:4 + H, 8:8 + W]                      # row stride W + 11 = 528 (4-byte aligned rows), offset base
        got = pp.unwarp_chunk_slices_backward(view, *a, 30, 200).cpu().numpy()
        assert np.array_equal(got, orc.unwarp_chunk_slices_backward(np.ascontiguousarray(view.cpu().numpy()), *a, 30, 200, **kernel_oracle(orc, "f64lerp")))
        for dt in ("uint16", "int16", "uint8", "int8", "int32", "uint32", "float64"):
            v = typed_image(dt, (D, H, W + 3), 600 + len(dt))        # W + 3 = 520: rows are 4-byte aligned for every type
            got = pp.unwarp_chunk_slices_backward(torch.from_numpy(v).cuda(), *a, 3, 280).cpu().numpy()
            bits = {1: "8-bit", 2: "16-bit", 4: "32-bit", 8: "float64"}[np.dtype(dt).itemsize]
            assert hip.last_kernel().startswith("stack_wg_kernel<NF=4,scipy," + bits), (dt, hip.last_kernel())
            assert got.dtype == np.dtype(dt) and np.array_equal(got, orc.unwarp_chunk_slices_backward(v, *a, 3, 280, poly=orc.POLY_KERNEL)), dt
    finally:
        hip.set_option("x_stack_wg", 1)


def test_stack_wg_kernel_xcd_runs_tile_order(hip, orc):
    """Round 4: stack_wg_kernel deals its (tile column, tile row, depth chunk) triples to the eight XCDs in contiguous runs (the
    default for integer stacks, option xcd_remap = 1 for float32 ones) or by whole tile rows (the default for float32 stacks whose
    tile rows pad to a multiple of eight cheaply).  A permutation of the workgroups only: every voxel equal
    to the grid order's and to the oracle, on grids whose workgroup count does and does not divide by eight, with ragged tiles
    and a ragged last depth chunk."""
    torch = pytest.importorskip("torch")
    a = (250.3, 140.8, [1.0, 3.0e-5, -4.0e-8, 1e-11, -2e-14])
    hip.set_option("x_stack_wg", 2)
    try:
        # (the last two: 15 and 16 tile rows -- float32 stacks then go by tile rows, the first with a vacant sixteenth row)
        for (D, H, W, r0, r1) in ((21, 300, 517, 7, 291), (9, 200, 640, 0, 199), (5, 97, 130, 3, 60), (6, 520, 300, 5, 474), (5, 530, 200, 0, 511)):
            vol = noise(700 + D, (D, H, W))
            want = orc.unwarp_chunk_slices_backward(vol, *a, r0, r1, **kernel_oracle(orc, "f64lerp"))
            for order in (1, 0, 2):
                hip.set_option("x_xcd_remap", order)
                got = pp.unwarp_chunk_slices_backward(torch.from_numpy(vol).cuda(), *a, r0, r1).cpu().numpy()
                assert hip.last_kernel() == "stack_wg_kernel<NF=5,f64lerp>", hip.last_kernel()
                assert np.array_equal(got, want), (D, H, W, order)
            v = typed_image("uint16", (D, H, W + (4 - W % 4) % 4), 710 + D)
            want = orc.unwarp_chunk_slices_backward(v, *a, r0, r1, poly=orc.POLY_KERNEL)
            for order in (2, 0):
                hip.set_option("x_xcd_remap", order)
                got = pp.unwarp_chunk_slices_backward(torch.from_numpy(v).cuda(), *a, r0, r1).cpu().numpy()
                assert hip.last_kernel().startswith("stack_wg_kernel<NF=5,scipy,16-bit"), hip.last_kernel()
                assert np.array_equal(got, want), (D, H, W, order)
    finally:
        hip.set_option("x_xcd_remap", 2)
        hip.set_option("x_stack_wg", 1)


def test_stack_wg_integer_streams_under_the_blend(hip, orc):
    """Round 4: the integer stack kernel issues the LDS-DMA of projection d + 1 from an asm statement the compiler does not track, blends
    projection d meanwhile, and at the top of the next projection waits with `s_waitcnt vmcnt(16)` -- for the fill only, the sixteen
    stores of the blend stay in flight (vector memory operations of a wave complete in issue order).  A wait that let a blend start
    before its slab had landed would show as voxels that differ from the oracle or from run to run: a shard of many workgroups and
    projections, every voxel against the oracle once and 40 more launches against that result; ragged waves (rows not a multiple of 16,
    width not a multiple of 64) take the full wait and are part of it."""
    torch = pytest.importorskip("torch")
    dot05 = [1.00227490554, -2.99523692178e-05, 8.99519088e-08, -1.57066461911e-10, 8.08880211618e-14]      # BASELINE cfg1, 1280 wide
    # (frames of the calibration's own aspect ratio: beyond it the model is not monotone and the call takes the checked kernels)
    for dt, (D, H, W), (r0, r1) in (("uint16", (37, 560, 900), (0, 559)), ("uint8", (20, 400, 644), (9, 390))):
        sc = 1280.0 / W
        a = (588.692801577 / sc, 462.092631791 / sc, [c * sc ** i for i, c in enumerate(dot05)])
        v = typed_image(dt, (D, H, W), 900 + D)
        dev = torch.from_numpy(v).cuda()
        hip.set_option("x_stack_wg", 2)                         # also for launches this small
        try:
            first = pp.unwarp_chunk_slices_backward(dev, *a, r0, r1).cpu().numpy()
            assert hip.last_kernel().startswith("stack_wg_kernel<NF=5,scipy,"), hip.last_kernel()
            assert np.array_equal(first, orc.unwarp_chunk_slices_backward(v, *a, r0, r1, poly=orc.POLY_KERNEL)), dt
            for rep in range(40):
                again = pp.unwarp_chunk_slices_backward(dev, *a, r0, r1)
                assert torch.equal(again, torch.from_numpy(first).cuda()), (dt, rep)
            hip.set_option("x_store_wait", 0)
            assert np.array_equal(pp.unwarp_chunk_slices_backward(dev, *a, r0, r1).cpu().numpy(), first), dt
        finally:
            hip.set_option("x_store_wait", 1)
            hip.set_option("x_stack_wg", 1)


def test_spline_gather_tiles_with_the_tallest_boxes(hip, orc):
    """Round 4's large-frame campaign (tools/fuzz_parity.py, FUZZ_BIG) found spline_wg_kernel's last LDS-DMA load of a box of
    full height writing its trailing lanes behind the slab, into the row tables: the tiles with the tallest boxes came out with
    some rows' coordinates zeroed, differently from run to run (a latent bug of the earlier rounds: their campaigns reached that
    kernel only with small distortions).  This is the campaign's case: a strong model, orders 3-5, repeated."""
    h, w = 1571, 1532
    img = (np.random.default_rng(5).random((h, w)) * 400.0 - 100.0).astype(np.float32)
    xc, yc, fact = 499.99635858988756, 218.84785084205987, [1.0, 1.9458361635865997e-05, 6.617751424219899e-09, 8.312873297400471e-13]
    for order in (3, 4, 5):
        want = orc.unwarp_image_backward(img, xc, yc, fact, order=order, mode="reflect", poly=orc.POLY_KERNEL)
        for rep in range(3):
            for blend in (None, "scipy"):
                got = pp.unwarp_image_backward(img, xc, yc, fact, order=order, mode="reflect", blend=blend)
                assert "spline_wg_kernel" in hip.last_kernel()
                bad = np.count_nonzero(np.abs(got.astype(np.float64) - want) > 1e-4)
                assert bad == 0, (order, rep, blend, bad)
                assert np.count_nonzero(got != want) <= 16, (order, rep, blend)      # (one-ulp pixels of the 2^-64 restarts / the factorised sum)
