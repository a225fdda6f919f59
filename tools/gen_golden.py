#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (discorpy 1.7.0 at /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Each fixture
holds the inputs (or the seed that regenerates them) and the reference's outputs; nothing of the
reference's source is stored.  SURVEY.md section 8(c) lists the cases G1..G9.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy.ndimage as ndi  # noqa: E402
from scipy.ndimage import map_coordinates  # noqa: E402
import discorpy.post.postprocessing as post  # noqa: E402
import discorpy.proc.processing as proc  # noqa: E402
from discorpy_amd import configs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    if os.path.exists(path):
        old = np.load(path)
        if sorted(old.files) == sorted(arrays) and all(
                np.asarray(arrays[k]).dtype == old[k].dtype and np.array_equal(old[k], arrays[k], equal_nan=True)
                for k in arrays):
            print("%-28s unchanged" % (name + ".npz"))
            return
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def f64(v):
    return np.asarray(v, dtype=np.float64)


# ---- G1: the reference's own unwarp_image_backward test input (tests/test_postprocessing.py:77-85)
hei = wid = 64
mat = np.zeros((hei, wid), dtype=np.float32)
mat[4:-3, 4:-3] = 1.0
fact = [1.0, 3.0 * 10 ** (-3)]
save("g1_box64", mat=mat, xcenter=f64(wid // 2), ycenter=f64(hei // 2), list_fact=f64(fact),
     out_order1=post.unwarp_image_backward(mat, wid // 2, hei // 2, fact),
     out_order0=post.unwarp_image_backward(mat, wid // 2, hei // 2, fact, order=0))

# ---- G2: the reference's slice / chunk test volume (tests/test_postprocessing.py:98-123)
mat = np.zeros((hei, wid), dtype=np.float32)
mat[:, 6:-8:8] = 1.0
mat = np.float32(ndi.binary_dilation(np.int16(mat), iterations=1))
mat3d = np.zeros((10, hei, wid), dtype=np.float32)
mat3d[:] = mat
x0, y0 = wid // 2, hei // 2
save("g2_stripes10x64x64", mat=mat, depth=np.int64(10), xcenter=f64(x0), ycenter=f64(y0), list_fact=f64(fact),
     index=np.int64(y0), slice_out=post.unwarp_slice_backward(mat3d, x0, y0, fact, y0),
     start=np.int64(y0 - 5), stop=np.int64(y0 + 5),
     chunk_out=post.unwarp_chunk_slices_backward(mat3d, x0, y0, fact, y0 - 5, y0 + 5))

# ---- G3: the reference's perspective test (tests/test_postprocessing.py:205-238)
hor_line1 = np.asarray([[10.0 + 2 * i / 32, i] for i in range(64)])
hor_line2 = np.asarray([[60.0 - 5 * i / 32, i] for i in range(64)])
ver_line1 = np.asarray([[i, 5.0 + 3 * i / 32] for i in range(64)])
ver_line2 = np.asarray([[i, 60.0 - 3 * i / 32] for i in range(64)])
src_pts, tgt_pts = proc.generate_source_target_perspective_points(
    [hor_line1, hor_line2], [ver_line1, ver_line2], equal_dist=False, scale="mean", optimizing=False)
pers_f = proc.calc_perspective_coefficients(src_pts, tgt_pts, mapping="forward")
pers_b = proc.calc_perspective_coefficients(src_pts, tgt_pts, mapping="backward")
mat = np.zeros((64, 64), dtype=np.float32)
mat[10:11] = np.float32(1.0)
mat[45:46] = np.float32(0.5)
cor1 = post.correct_perspective_image(mat, pers_b)
cor2 = post.correct_perspective_image(cor1, pers_f)
save("g3_perspective64", mat=mat, coef_backward=f64(pers_b), coef_forward=f64(pers_f), cor_backward=cor1,
     cor_forward_of_backward=cor2, cor_backward_order0=post.correct_perspective_image(mat, pers_b, order=0))

# ---- G4: config 1 -- data/dot_pattern_05.jpg + data/coef_dot_05.txt.  The decoded JPEG is a data
# file of the reference; a 160x192 crop around the centre of distortion and eight full output rows
# are kept (JPEG decoding is library dependent: never decode on the GPU box).
from PIL import Image  # noqa: E402
img = np.array(Image.open("/root/reference/data/dot_pattern_05.jpg"), dtype=np.float32)
assert img.shape == configs.DOT_05_SHAPE
with open("/root/reference/data/coef_dot_05.txt") as fh:
    vals = [float(line.split()[-1]) for line in fh if line.strip()]
xc, yc, coef = vals[0], vals[1], vals[2:]
assert abs(xc - configs.XCENTER_DOT_05) < 1e-9 and len(coef) == 5
full = post.unwarp_image_backward(img, xc, yc, coef)
rows = np.array([0, 10, 137, 400, 462, 463, 700, 799])
crop = (slice(384, 544), slice(492, 684))      # 160 x 192 window holding the centre (462, 588)
crop_in = np.ascontiguousarray(img[crop])
# the same call on the crop alone (centre shifted): a self-contained input/output pair
crop_out = post.unwarp_image_backward(crop_in, xc - 492, yc - 384, coef)
save("g4_dot_pattern_05", crop_in=crop_in, crop_xcenter=f64(xc - 492), crop_ycenter=f64(yc - 384),
     list_fact=f64(coef), crop_out=crop_out, full_rows=rows, full_out_rows=full[rows],
     full_stats=f64([img.min(), img.max(), img.mean(), full.mean(), full[400, 640], full[10, 10], full[799, 1279]]),
     full_in_rows_band=img[395:406])

# ---- G4b: config 1 at full size -- the decoded 800 x 1280 frame as uint8 (mode L: the decode IS integers, so this
# is the float32 image the reference loads, losslessly), the SHA-256 of the reference's full float32 output and a
# 25 x 40 lattice of its pixels.  (The 4 MB output itself is not stored.)
import hashlib  # noqa: E402
assert np.array_equal(img, img.astype(np.uint8).astype(np.float32))
lat = (slice(0, 800, 32), slice(0, 1280, 32))
save("g4b_dot_pattern_05_full", frame_u8=img.astype(np.uint8), xcenter=f64(xc), ycenter=f64(yc), list_fact=f64(coef),
     out_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(full).tobytes()).digest(), dtype=np.uint8),
     out_lattice=np.ascontiguousarray(full[lat]), out_order0_sha256=np.frombuffer(hashlib.sha256(
         np.ascontiguousarray(post.unwarp_image_backward(img, xc, yc, coef, order=0)).tobytes()).digest(), dtype=np.uint8))

# ---- G5: configs 2 / 5 at reduced size, seeded noise; float32 coordinate planes kept
def coords_ref(h, w, xc, yc, fact):
    xu = np.arange(w) - xc
    yu = np.arange(h) - yc
    xm, ym = np.meshgrid(xu, yu)
    ru = np.sqrt(xm ** 2 + ym ** 2)
    fm = np.sum(np.asarray([f * ru ** i for i, f in enumerate(fact)]), axis=0)
    return (np.float32(np.clip(yc + fm * ym, 0, h - 1)), np.float32(np.clip(xc + fm * xm, 0, w - 1)))


for name, (h, w), seed, (xc5, yc5, fact5) in [
        ("g5_cfg2_160", (160, 160), 11, configs.rescale_model(160)),
        ("g5_offcentre_150x200", (150, 200), 12, (configs.rescale_model(200)[0] + 60.25,
                                                  configs.rescale_model(200)[1] - 37.5,
                                                  configs.rescale_model(200)[2])),
        ("g5_cfg5_9term_144", (144, 144), 13, (72.3, 77.7,
                                               [a * (8192 / 144) ** i for i, a in enumerate(configs.CFG5_FACT)]))]:
    im = np.random.default_rng(seed).random((h, w), dtype=np.float32)
    yd, xd = coords_ref(h, w, xc5, yc5, fact5)
    save(name, seed=np.int64(seed), shape=np.array([h, w]), xcenter=f64(xc5), ycenter=f64(yc5), list_fact=f64(fact5),
         yd=yd, xd=xd, out_order1=post.unwarp_image_backward(im, xc5, yc5, fact5),
         out_order0=post.unwarp_image_backward(im, xc5, yc5, fact5, order=0))

# ---- G6: slice / chunk on a (6, 800, 1280)-shaped problem, coef_dot_05, rows 0, 14, 400, 799
vol = np.random.default_rng(21).random((3, 800, 1280), dtype=np.float32)
c = configs
g6 = dict(seed=np.int64(21), shape=np.array(vol.shape), xcenter=f64(c.XCENTER_DOT_05), ycenter=f64(c.YCENTER_DOT_05),
          list_fact=f64(c.COEF_DOT_05), rows=np.array([0, 14, 400, 799]))
for r in (0, 14, 400, 799):
    g6["slice_%d" % r] = post.unwarp_slice_backward(vol, c.XCENTER_DOT_05, c.YCENTER_DOT_05, c.COEF_DOT_05, r)
g6["slice_frac_400p5"] = post.unwarp_slice_backward(vol, c.XCENTER_DOT_05, c.YCENTER_DOT_05, c.COEF_DOT_05, 400.5)
g6["chunk_395_402"] = post.unwarp_chunk_slices_backward(vol, c.XCENTER_DOT_05, c.YCENTER_DOT_05, c.COEF_DOT_05, 395, 402)
g6["chunk_0_2"] = post.unwarp_chunk_slices_backward(vol, c.XCENTER_DOT_05, c.YCENTER_DOT_05, c.COEF_DOT_05, 0, 2)
g6["chunk_797_799"] = post.unwarp_chunk_slices_backward(vol, c.XCENTER_DOT_05, c.YCENTER_DOT_05, c.COEF_DOT_05, 797, 799)
save("g6_stack3x800x1280", **g6)

# ---- G7: fused perspective -> radial, one map_coordinates call at the composed coordinates
h = w = 144
im = np.random.default_rng(31).random((h, w), dtype=np.float32)
s = 4096 / w
coef7 = list(configs.CFG3_COEF)
# rescale the 4096^2 homography to a 256^2 frame: x' = x/s  =>  c3,c6 / s ; c7,c8 * s
coef7 = [coef7[0], coef7[1], coef7[2] / s, coef7[3], coef7[4], coef7[5] / s, coef7[6] * s, coef7[7] * s]
xc7, yc7, fact7 = configs.rescale_model(w)
yp, xp = post._generate_perspective_map(im, coef7)
xp = np.float64(xp.reshape(h, w))
yp = np.float64(yp.reshape(h, w))
xu = xp - xc7
yu = yp - yc7
ru = np.sqrt(xu ** 2 + yu ** 2)
fm = np.sum(np.asarray([f * ru ** i for i, f in enumerate(fact7)]), axis=0)
xd = np.float32(np.clip(xc7 + fm * xu, 0, w - 1))
yd = np.float32(np.clip(yc7 + fm * yu, 0, h - 1))
fused = map_coordinates(im, (yd.reshape(-1, 1), xd.reshape(-1, 1)), order=1, mode="reflect").reshape(h, w)
twopass = post.correct_perspective_image(post.unwarp_image_backward(im, xc7, yc7, fact7), coef7)
save("g7_fused144", seed=np.int64(31), shape=np.array([h, w]), xcenter=f64(xc7), ycenter=f64(yc7), list_fact=f64(fact7),
     list_coef=f64(coef7), fused_out=fused, twopass_out=twopass, yd=yd, xd=xd,
     persp_out=post.correct_perspective_image(im, coef7))

# ---- G8: clipping stress -- 55 % of the coordinates clipped; all eight modes must agree
h, w = 120, 180
im = np.random.default_rng(41).random((h, w), dtype=np.float32)
fact8 = [1.3, 2e-3]
outs = {}
for order in (0, 1):
    base = post.unwarp_image_backward(im, 90.0, 60.0, fact8, order=order, mode="reflect")
    for mode in ("grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        assert np.array_equal(base, post.unwarp_image_backward(im, 90.0, 60.0, fact8, order=order, mode=mode)), mode
    outs["out_order%d" % order] = base
yd, xd = coords_ref(h, w, 90.0, 60.0, fact8)
save("g8_clip120x180", seed=np.int64(41), shape=np.array([h, w]), xcenter=f64(90.0), ycenter=f64(60.0),
     list_fact=f64(fact8), clipped_fraction=f64(np.mean((xd == 0) | (xd == w - 1) | (yd == 0) | (yd == h - 1))), **outs)

# ---- G9: explicit coordinates incl. half-integers (order 0 rounds half up), edges, tiny fractions
im = np.random.default_rng(51).random((33, 47), dtype=np.float32)
ys = np.array([0.0, 0.5, 1.5, 2.5, 31.5, 32.0, 15.25, 7.999999, 1e-30, 31.999998, 0.49999997, 12.5, 20.0, 3.5],
              dtype=np.float32)
xs = np.array([0.0, 0.5, 2.5, 45.5, 46.0, 46.0, 22.75, 1e-20, 0.49999997, 45.999996, 10.5, 0.0, 46.0, 44.5],
              dtype=np.float32)
rng = np.random.default_rng(52)
ys = np.concatenate([ys, (rng.random(500) * 32).astype(np.float32)])
xs = np.concatenate([xs, (rng.random(500) * 46).astype(np.float32)])
yd64 = rng.random(300) * 32
xd64 = rng.random(300) * 46
save("g9_points33x47", seed=np.int64(51), shape=np.array([33, 47]), ys=ys, xs=xs,
     out_order0=map_coordinates(im, (ys, xs), order=0, mode="reflect"),
     out_order1=map_coordinates(im, (ys, xs), order=1, mode="reflect"),
     ys64=yd64, xs64=xd64, out64_order1=map_coordinates(im, (yd64, xd64), order=1, mode="reflect"),
     out64_order0=map_coordinates(im, (yd64, xd64), order=0, mode="reflect"))
# ---- G10: util.unwarp_color_image_backward (utility.py:278-342): channels, explicit pads, pad modes
import discorpy.util.utility as util  # noqa: E402
rgb = np.random.default_rng(61).random((40, 56, 3), dtype=np.float32)
fact10 = [1.0, 4e-3, 2e-5]
g10 = dict(seed=np.int64(61), shape=np.array(rgb.shape), xcenter=f64(27.4), ycenter=f64(19.1), list_fact=f64(fact10))
g10["nopad"] = util.unwarp_color_image_backward(rgb, 27.4, 19.1, fact10)
g10["pad_3_5_2_7_edge"] = util.unwarp_color_image_backward(rgb, 27.4, 19.1, fact10, pad=(3, 5, 2, 7), pad_mode="edge")
g10["pad_4_constant"] = util.unwarp_color_image_backward(rgb, 27.4, 19.1, fact10, pad=4)
g10["pad_4_reflect_order0"] = util.unwarp_color_image_backward(rgb, 27.4, 19.1, fact10, order=0, pad=4, pad_mode="reflect")
g10["gray_pad_6_mean"] = util.unwarp_color_image_backward(rgb[:, :, 1], 27.4, 19.1, fact10, pad=6, pad_mode="mean")
save("g10_color40x56x3", **g10)
# ---- G11: spline orders 2..5 (map_coordinates with prefilter), every boundary mode
im = np.random.default_rng(71).random((45, 60), dtype=np.float32)
fact11 = [1.0, 3e-3, 2e-5]
coef11 = [0.97, -0.02, 2.0, 0.015, 0.95, 1.5, -2e-4, 3e-4]
g11 = dict(seed=np.int64(71), shape=np.array(im.shape), xcenter=f64(31.3), ycenter=f64(21.8), list_fact=f64(fact11),
           list_coef=f64(coef11))
MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")
for order in (2, 3, 4, 5):
    for mode in MODES:
        g11["radial_o%d_%s" % (order, mode)] = post.unwarp_image_backward(im, 31.3, 21.8, fact11, order=order, mode=mode)
for mode in MODES:
    g11["persp_o3_%s" % mode] = post.correct_perspective_image(im, coef11, order=3, mode=mode)   # demo_07.py:60 uses order=3
pts_y = (np.random.default_rng(72).random(500) * 44).astype(np.float32)
pts_x = (np.random.default_rng(73).random(500) * 59).astype(np.float32)
g11["pts_y"], g11["pts_x"] = pts_y, pts_x
for order in (2, 3, 4, 5):
    g11["points_o%d_reflect" % order] = map_coordinates(im, (pts_y, pts_x), order=order, mode="reflect")
save("g11_spline45x60", **g11)
# ---- G12: element types other than float32 (output dtype = input dtype; unwarp_slice_backward -> float32)
def typed_image(dt, shape, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.random(shape) * 2000.0 - 700.0).astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=np.int64).astype(dt)


g12 = dict(shape=np.array([40, 52]), xcenter=f64(24.6), ycenter=f64(20.3), list_fact=f64([1.0, 5e-3, 3e-5]),
           list_coef=f64([0.96, -0.03, 1.5, 0.02, 0.97, 1.0, -3e-4, 2e-4]), vol_shape=np.array([5, 40, 52]),
           index=np.int64(17), start=np.int64(8), stop=np.int64(21))
pts_y = (np.random.default_rng(82).random(400) * 39).astype(np.float32)
pts_x = (np.random.default_rng(83).random(400) * 51).astype(np.float32)
# exact halves between neighbouring integers: the integer stores round half away from zero
pts_y[:40] = np.repeat(np.arange(8, dtype=np.float32), 5)
pts_x[:40] = np.tile(np.array([0.5, 1.5, 2.5, 3.5, 4.0], dtype=np.float32), 8)
g12["pts_y"], g12["pts_x"] = pts_y, pts_x
for k, dt in enumerate(("uint8", "int8", "uint16", "int16", "uint32", "int32", "float64")):
    im = typed_image(dt, (40, 52), 800 + k)
    im[:8, :5] = np.arange(40).reshape(8, 5).astype(im.dtype) - (20 if np.dtype(dt).kind != "u" else 0)
    g12["seed_" + dt] = np.int64(800 + k)
    for order in (0, 1, 3):
        g12["radial_o%d_%s" % (order, dt)] = post.unwarp_image_backward(im, 24.6, 20.3, g12["list_fact"], order=order)
        g12["points_o%d_%s" % (order, dt)] = map_coordinates(im, (pts_y, pts_x), order=order, mode="reflect")
    g12["radial_o2_nearest_" + dt] = post.unwarp_image_backward(im, 24.6, 20.3, g12["list_fact"], order=2, mode="nearest")
    g12["persp_o1_" + dt] = post.correct_perspective_image(im, g12["list_coef"])
    g12["persp_o5_wrap_" + dt] = post.correct_perspective_image(im, g12["list_coef"], order=5, mode="grid-wrap")
    vol = typed_image(dt, (5, 40, 52), 900 + k)
    g12["slice_" + dt] = post.unwarp_slice_backward(vol, 24.6, 20.3, g12["list_fact"], 17)
    g12["chunk_" + dt] = post.unwarp_chunk_slices_backward(vol, 24.6, 20.3, g12["list_fact"], 8, 21)
    assert g12["radial_o1_" + dt].dtype == np.dtype(dt) and g12["chunk_" + dt].dtype == np.dtype(dt)
    assert g12["slice_" + dt].dtype == np.float32
save("g12_dtypes40x52", **g12)
# ---- G12b: the element types scipy reads as doubles with loss or stores through an undefined cast: int64, uint64 (taps above
# 2^53 round on the way in; a result of 2^63 / 2^64 -- any blend of taps at the type's maximum -- is stored as the x86-64
# cvttsd2si sequence stores it: INT64_MIN / 0) and bool (stored by truncating the double).  Inputs: conftest.wide_image.
def wide_image(dt, shape, seed):
    rng = np.random.default_rng(seed)
    if np.dtype(dt) == np.bool_:
        return rng.random(shape) < 0.5
    info = np.iinfo(dt)
    im = rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=dt)
    flat = im.reshape(-1)
    flat[::7] = rng.integers(0, 1 << 20, size=flat[::7].shape).astype(dt)           # small values between the huge ones
    flat[3::11] = info.max
    flat[5::13] = info.min
    return im


g12b = dict(shape=np.array([40, 52]), xcenter=f64(24.6), ycenter=f64(20.3), list_fact=f64([1.0, 5e-3, 3e-5]),
            list_coef=g12["list_coef"], vol_shape=np.array([5, 40, 52]), index=np.int64(17), start=np.int64(8), stop=np.int64(21),
            pts_y=pts_y, pts_x=pts_x)
for k, dt in enumerate(("int64", "uint64", "bool")):
    im = wide_image(dt, (40, 52), 850 + k)
    g12b["seed_" + dt] = np.int64(850 + k)
    for order in (0, 1, 3):
        g12b["radial_o%d_%s" % (order, dt)] = post.unwarp_image_backward(im, 24.6, 20.3, g12b["list_fact"], order=order)
        g12b["points_o%d_%s" % (order, dt)] = map_coordinates(im, (pts_y, pts_x), order=order, mode="reflect")
    g12b["persp_o1_" + dt] = post.correct_perspective_image(im, g12b["list_coef"])
    vol = wide_image(dt, (5, 40, 52), 950 + k)
    g12b["slice_" + dt] = post.unwarp_slice_backward(vol, 24.6, 20.3, g12b["list_fact"], 17)
    g12b["chunk_" + dt] = post.unwarp_chunk_slices_backward(vol, 24.6, 20.3, g12b["list_fact"], 8, 21)
    assert g12b["radial_o1_" + dt].dtype == np.dtype(dt) and g12b["chunk_" + dt].dtype == np.dtype(dt) and g12b["slice_" + dt].dtype == np.float32
# complex data: scipy interpolates the real and the imaginary part separately
cim = (np.random.default_rng(861).random((40, 52)) + 1j * np.random.default_rng(862).random((40, 52))).astype(np.complex64)
g12b["radial_o1_complex64"] = post.unwarp_image_backward(cim, 24.6, 20.3, g12b["list_fact"])
g12b["persp_o1_complex64"] = post.correct_perspective_image(cim, g12b["list_coef"])
save("g12b_wide_types40x52", **g12b)
# ---- G13: unwarp_line_forward (postprocessing.py:36-64) on eight jittered dot lines of the 800 x 1280 pattern
rng13 = np.random.default_rng(5)
lines13 = [np.column_stack([np.full(12, 40.0 * k) + rng13.uniform(-2, 2, 12), np.linspace(5, 1270, 12) + rng13.uniform(-1, 1, 12)])
           for k in range(1, 9)]
fact13 = np.asarray([1.0, -2.9e-5, 8.9e-8, -1.5e-10, 8.0e-14])
save("g13_lines_forward", lines=np.asarray(lines13), xcenter=f64(588.69), ycenter=f64(462.09), list_fact=fact13,
     out=np.asarray(post.unwarp_line_forward(lines13, 588.69, 462.09, fact13)))
# ---- G19: correct_perspective_line (postprocessing.py:414-441) on the reference's own test lines (tests/test_postprocessing.py:
# 159-176): forward coefficients applied to the lines, backward coefficients applied to the result
hor19 = [np.asarray([[10.0 + 2 * i / 32, i] for i in range(32)]), np.asarray([[26.0 - 5 * i / 32, i] for i in range(32)])]
ver19 = [np.asarray([[i, 0.0 + 3 * i / 32] for i in range(32)]), np.asarray([[i, 20.0 - 3 * i / 32] for i in range(32)])]
src19, tgt19 = proc.generate_source_target_perspective_points(hor19, ver19, equal_dist=False, scale="mean", optimizing=False)
fcoef19 = proc.calc_perspective_coefficients(src19, tgt19, mapping="forward")
bcoef19 = proc.calc_perspective_coefficients(src19, tgt19, mapping="backward")
fwd19 = post.correct_perspective_line(hor19 + ver19, fcoef19)
save("g19_perspective_lines", lines=np.asarray(hor19 + ver19), fcoef=f64(fcoef19), bcoef=f64(bcoef19), forward=np.asarray(fwd19),
     back=np.asarray(post.correct_perspective_line(fwd19, bcoef19)))
# ---- G14: pad=True of util.unwarp_color_image_backward (utility.py:238-263): automatic pad widths from the forward
# model fitted by proc.transform_coef_backward_and_forward (processing.py:615-674), then the padded unwarp; plus the
# reference's own pad=True case (tests/test_utility.py:92-101) and the fitted coefficients themselves
rgb14 = np.random.default_rng(141).random((40, 56, 3), dtype=np.float32)
fact14 = [1.0, -6e-3, -2e-5]
g14 = dict(seed=np.int64(141), shape=np.array(rgb14.shape), xcenter=f64(27.4), ycenter=f64(19.1), list_fact=f64(fact14))
g14["pads"] = np.asarray(util._calc_pad(True, 40, 56, 27.4, 19.1, fact14), dtype=np.int64)
g14["pad_true_constant"] = util.unwarp_color_image_backward(rgb14, 27.4, 19.1, fact14, pad=True)
g14["pad_true_edge_order0"] = util.unwarp_color_image_backward(rgb14, 27.4, 19.1, fact14, order=0, pad=True, pad_mode="edge")
g14["gray_pad_true_reflect"] = util.unwarp_color_image_backward(rgb14[:, :, 2], 27.4, 19.1, fact14, pad=True, pad_mode="reflect")
grid14 = [[gy - 19.1, gx - 27.4] for gy in np.linspace(0, 40, 40) for gx in np.linspace(0, 56, 40)]
g14["forward_fact"] = f64(proc.transform_coef_backward_and_forward(fact14, ref_points=grid14))
g14["default_grid_backward"] = f64(proc.transform_coef_backward_and_forward([1.0, -3e-5, 9e-8]))
g14["default_grid_forward"] = f64(proc.transform_coef_backward_and_forward([1.0, -3e-5, 9e-8], mapping="forward"))
box = np.ones((40, 60), dtype=np.float32)      # tests/test_utility.py:75-101
box[:5] = 0
box[-5:] = 0
box[:, :5] = 0
box[:, -5:] = 0
ref14 = [[i - 20.0, j - 30.0] for i in range(0, 40, 10) for j in range(0, 60, 10)]
tfact14 = proc.transform_coef_backward_and_forward([1.0, 0.1, 0.01], ref_points=ref14)
g14["reftest_tfact"] = f64(tfact14)
g14["reftest_pads"] = np.asarray(util._calc_pad(True, 40, 60, 30.0, 20.0, tfact14), dtype=np.int64)
g14["reftest_pad_true"] = util.unwarp_color_image_backward(box, 30.0, 20.0, tfact14, 1, "constant", True, "constant")
save("g14_autopad40x56x3", **g14)

# G15: a FOLDING radial model on a chunk of rows -- the one documented deviation of the stack path that the reference's
# own tests never reach.  The reference crops the band [yd_min, yd_max) from the first row's minimum and the last row's
# maximum (postprocessing.py:289-301) and lets map_coordinates reflect inside that band; rows whose coordinates leave
# the band (possible only for a non-monotone map) therefore read reflected samples.  discorpy_amd samples the
# projection at the absolute coordinate.  The fixture pins both: the reference's output, the mask of pixels whose row
# coordinate leaves the band, and the absolute-coordinate result (scipy on the whole projection, same float32 coordinates).
def g15():
    d, h, w = 4, 64, 96
    xc, yc, fact = 48.3, 30.2, [1.0, 0.0, -4.0e-4]
    start, stop = 2, 20
    vol = np.random.default_rng(1515).random((d, h, w), dtype=np.float32)
    ref = post.unwarp_chunk_slices_backward(vol, xc, yc, fact, start, stop)
    xu = np.arange(0, w) - xc
    yu = np.arange(start, stop + 1) - yc
    xu_mat, yu_mat = np.meshgrid(xu, yu)
    ru = np.sqrt(xu_mat ** 2 + yu_mat ** 2)
    fm = np.sum(np.asarray([f * ru ** i for i, f in enumerate(fact)]), axis=0)
    xd = np.float32(np.clip(xc + fm * xu_mat, 0, w - 1))
    yd = np.float32(np.clip(yc + fm * yu_mat, 0, h - 1))
    yd_min = int(np.floor(np.amin(yd[0].astype(np.float64))))        # as the reference: first row's minimum,
    yd_max = int(np.ceil(np.amax(yd[-1].astype(np.float64)))) + 1    # last row's maximum (float64 there; same integers here)
    outside = (yd < yd_min) | (yd > yd_max - 1)
    absolute = np.asarray([map_coordinates(vol[i], (yd, xd), order=1, mode="reflect") for i in range(d)])
    assert outside.any() and not outside.all()
    assert np.array_equal(ref[:, ~outside], absolute[:, ~outside])
    assert not np.array_equal(ref[:, outside], absolute[:, outside])
    # the same chunk of a float64 and a uint16 stack (output dtype = input dtype): rows BELOW the band get a band-relative
    # coordinate larger in magnitude than the absolute one, where the reference's float32 subtraction yd_mat - yd_min rounds
    ref_f64 = post.unwarp_chunk_slices_backward(vol.astype(np.float64), xc, yc, fact, start, stop)
    ref_u16 = post.unwarp_chunk_slices_backward((vol * 60000).astype(np.uint16), xc, yc, fact, start, stop)
    # a second geometry whose rows leave the band on the low side by more than the coordinate itself
    vol2 = np.random.default_rng(1516).random((2, 55, 58), dtype=np.float32)
    case2 = (29.3, 8.4, [0.9105341112829652, -0.011490848185174918, -0.0006863296016646205], 46, 52)
    ref2 = post.unwarp_chunk_slices_backward(vol2, *case2)
    ref2_f64 = post.unwarp_chunk_slices_backward(vol2.astype(np.float64), *case2)
    save("g15_folding_chunk", seed=np.int64(1515), shape=np.array([d, h, w]), xcenter=f64(xc), ycenter=f64(yc),
         list_fact=f64(fact), start=np.int64(start), stop=np.int64(stop), ref_out=ref, outside_band=outside,
         absolute_out=absolute.astype(np.float32), band=np.array([yd_min, yd_max]), ref_out_f64=ref_f64, ref_out_u16=ref_u16,
         case2_seed=np.int64(1516), case2_shape=np.array([2, 55, 58]), case2_xcenter=f64(case2[0]), case2_ycenter=f64(case2[1]),
         case2_list_fact=f64(case2[2]), case2_rows=np.array(case2[3:]), case2_ref_out=ref2, case2_ref_out_f64=ref2_f64)


g15()


# G16: correct_perspective_image with a caller's map_index whose coordinates leave the image, under every boundary mode,
# orders 0 and 1 -- the reference passes map_index and mode straight to scipy (postprocessing.py:489-491), so this pins
# the out-of-image handling (scipy's coordinate mapping, tap folding, cval = 0) to the reference's own call.
def g16():
    h, w = 23, 31
    rng = np.random.default_rng(1616)
    mat = rng.random((h, w), dtype=np.float32)
    n = h * w                      # (the reference reshapes the result to the image's shape, :492)
    ys = (rng.random(n) * h * 7 - h * 3).astype(np.float32)
    xs = (rng.random(n) * w * 7 - w * 3).astype(np.float32)
    ys[:50] = np.linspace(-2.0, h + 1.0, 50)
    xs[:50] = 11.25
    out = {}
    coef = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]      # unused once map_index is given
    for mode in ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap"):
        for order in (0, 1):
            out["%s_o%d" % (mode.replace("-", "_"), order)] = post.correct_perspective_image(mat, coef, order=order, mode=mode,
                                                                                             map_index=(ys, xs))
    save("g16_map_index_outside", seed=np.int64(1616), shape=np.array([h, w]), ys=ys, xs=xs, **out)


g16()


# G17: small random frames (2..69 pixels a side) through unwarp_image_backward and correct_perspective_image at a random
# order 0..5 and boundary mode -- the regime where the spline prefilter's boundary handling (and scipy's quirks in it) is
# visible in every pixel.  60 cases; inputs regenerate from the seed.
def g17_case(rng):
    h, w = int(rng.integers(2, 70)), int(rng.integers(2, 70))
    img = rng.random((h, w), dtype=np.float32)
    xc, yc = float(rng.uniform(0, w)), float(rng.uniform(0, h))
    fact = [1.0, float(rng.uniform(-2e-3, 2e-3)), float(rng.uniform(-3e-5, 3e-5))]
    order = int(rng.integers(0, 6))
    mode = MODES_ALL[int(rng.integers(0, 8))]
    coef = [1 + rng.uniform(-.05, .05), rng.uniform(-.05, .05), rng.uniform(-3, 3), rng.uniform(-.05, .05),
            1 + rng.uniform(-.05, .05), rng.uniform(-3, 3), rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4)]
    return img, xc, yc, fact, order, mode, [float(c) for c in coef]


MODES_ALL = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")


def g17():
    rng = np.random.default_rng(1717)
    out = {}
    for k in range(60):
        img, xc, yc, fact, order, mode, coef = g17_case(rng)
        out["radial_%02d" % k] = post.unwarp_image_backward(img, xc, yc, fact, order=order, mode=mode)
        out["persp_%02d" % k] = post.correct_perspective_image(img, coef, order=order, mode=mode)
    save("g17_small_frames_orders_modes", seed=np.int64(1717), ncases=np.int64(60), **out)


g17()


# G18: unwarp_slice_backward with an `index` the reference does not validate (:215) -- fractional, negative, past the last row --
# on float32, uint16 and float64 stacks.  40 cases; inputs regenerate from the seed.
def g18_case(rng, t):
    d, h, w = int(rng.integers(1, 3)), int(rng.integers(3, 70)), int(rng.integers(3, 70))
    dt = [np.float32, np.uint16, np.float64][t % 3]
    vol = rng.random((d, h, w))
    vol = (vol * 60000).astype(dt) if dt == np.uint16 else vol.astype(dt)
    xc, yc = float(rng.uniform(-0.2 * w, 1.2 * w)), float(rng.uniform(-0.2 * h, 1.2 * h))
    fact = [1.0 + float(rng.uniform(-.1, .1)), float(rng.uniform(-3e-3, 3e-3)), float(rng.uniform(-1e-4, 1e-4))]
    idx = [float(rng.uniform(-5, h + 5)), int(rng.integers(-3, h + 3)), float(rng.integers(0, h)) + 0.5][(t // 3) % 3]
    return vol, xc, yc, fact, idx


def g18():
    rng = np.random.default_rng(1818)
    out = {}
    for t in range(40):
        vol, xc, yc, fact, idx = g18_case(rng, t)
        out["slice_%02d" % t] = post.unwarp_slice_backward(vol, xc, yc, fact, idx)
    save("g18_slice_any_index", seed=np.int64(1818), ncases=np.int64(40), **out)


g18()
print("done")
