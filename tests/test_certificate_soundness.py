"""CPU suite: the host's tile-deviation certificate (csrc/api_core.cpp: tile_deviation_certified) against brute force.

The staged kernels take a 128 x 32 tile's source box from its corner pixels alone -- no per-pixel check -- whenever the
certificate says level 2.  Here the claim is checked the slow way, in NumPy, on random calibrations the certificate accepts:
every tap of every pixel of every tile must lie inside the box the kernels form (remap_wg_body / wg_corner_tap in
csrc/unwarp_kernels.hip).  Radial (reference postprocessing.py:138-145), perspective (:448-455) and -- new in round 5 -- the fused
map, whose box comes from the radial map at the corners of the tile's perspective bounding box and must also hold where the
inner clip cuts through a tile.  Needs the library (host arithmetic only), no GPU."""
import ctypes as C

import numpy as np
import pytest

from discorpy_amd import _ffi as F

TW, TH = 128, 32


def f32clip(v, hi):
    return np.clip(v.astype(np.float32), np.float32(0), np.float32(hi))


def radial(xp, yp, xc, yc, fact):
    xu, yu = xp - xc, yp - yc
    r = np.sqrt(xu * xu + yu * yu)
    b = np.zeros_like(r)
    for a in reversed(fact):
        b = b * r + a
    return xc + b * xu, yc + b * yu


def persp(x, y, c):
    den = c[6] * x + c[7] * y + 1.0
    return (c[0] * x + c[1] * y + c[2]) / den, (c[3] * x + c[4] * y + c[5]) / den


def level(kind, H, W, xc, yc, fact, coef):
    fa, nf = F.fact_array(fact) if fact is not None else (None, 0)
    ca = (C.c_double * 8)(*coef) if coef is not None else None
    return F.lib().dcp_debug_tile_certificate(kind, H, W, float(xc), float(yc), fa, nf, ca)


def check_boxes(kind, H, W, xc, yc, fact, coef):
    """Every tap of every pixel inside its tile's box; returns the number of tiles the inner clip cuts through (fused)."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    wmax, hmax = W - 1, H - 1
    if kind == F.MAP_RADIAL:
        xd, yd = radial(xs, ys, xc, yc, fact)
    elif kind == F.MAP_PERSPECTIVE:
        xd, yd = persp(xs, ys, coef)
    else:
        px, py = persp(xs, ys, coef)
        raw_x, raw_y = px, py
        px, py = f32clip(px, wmax).astype(np.float64), f32clip(py, hmax).astype(np.float64)
        xd, yd = radial(px, py, xc, yc, fact)
    xf, yf = f32clip(xd, wmax), f32clip(yd, hmax)
    xi = np.minimum(xf.astype(np.int64), W - 2)
    yi = np.minimum(yf.astype(np.int64), H - 2)
    cut = 0
    for ty in range(0, H, TH):
        for tx in range(0, W, TW):
            x1, y1 = min(tx + TW - 1, W - 1), min(ty + TH - 1, H - 1)
            cys, cxs = np.array([ty, ty, y1, y1]), np.array([tx, x1, tx, x1])
            if kind == F.MAP_FUSED:
                qx, qy = px[cys, cxs], py[cys, cxs]
                bx, by = radial(np.array([qx.min(), qx.max(), qx.min(), qx.max()]), np.array([qy.min(), qy.min(), qy.max(), qy.max()]), xc, yc, fact)
                cx, cy = f32clip(bx, wmax).astype(np.int64), f32clip(by, hmax).astype(np.int64)
                rx, ry = raw_x[ty:y1 + 1, tx:x1 + 1], raw_y[ty:y1 + 1, tx:x1 + 1]
                inside = (rx >= 0) & (rx <= wmax) & (ry >= 0) & (ry <= hmax)
                cut += int(inside.any() and not inside.all())
            else:
                cx, cy = xf[cys, cxs].astype(np.int64), yf[cys, cxs].astype(np.int64)
            bx0, bx1 = max(min(cx.min() - 1, W - 2), 0), min(cx.max() + 2, W - 1)
            by0, by1 = max(min(cy.min() - 1, H - 2), 0), min(cy.max() + 2, H - 1)
            txi, tyi = xi[ty:y1 + 1, tx:x1 + 1], yi[ty:y1 + 1, tx:x1 + 1]
            assert txi.min() >= bx0 and txi.max() + 1 <= bx1 and tyi.min() >= by0 and tyi.max() + 1 <= by1, (
                kind, (tx, ty), (bx0, bx1, by0, by1), (txi.min(), txi.max(), tyi.min(), tyi.max()))
    return cut


def mild_fact(rng, H, W, n):
    R = float(np.hypot(H, W))
    return [1.0 + float(rng.uniform(-0.03, 0.03))] + [float(rng.uniform(-0.05, 0.05)) / R ** i for i in range(1, n)]


def mild_coef(rng, H, W, s=0.05, p=4e-5, shift=0.12):
    return [1.0 + rng.uniform(-s, s), rng.uniform(-s, s), rng.uniform(-shift, shift) * W, rng.uniform(-s, s), 1.0 + rng.uniform(-s, s),
            rng.uniform(-shift, shift) * H, rng.uniform(-p, p), rng.uniform(-p, p)]


@pytest.mark.parametrize("seed", range(6))
def test_certified_radial_and_perspective_boxes_hold_every_tap(seed):
    rng = np.random.default_rng(100 + seed)
    H, W = int(rng.integers(200, 520)), int(rng.integers(300, 900))
    xc, yc = rng.uniform(0.1, 0.9) * W, rng.uniform(0.1, 0.9) * H
    done = 0
    for _ in range(12):
        fact = mild_fact(rng, H, W, int(rng.integers(2, 8)))
        if level(F.MAP_RADIAL, H, W, xc, yc, fact, None) >= 2:
            check_boxes(F.MAP_RADIAL, H, W, xc, yc, fact, None)
            done += 1
            break
    for _ in range(12):
        coef = mild_coef(rng, H, W)
        if level(F.MAP_PERSPECTIVE, H, W, 0.0, 0.0, None, coef) >= 2:
            check_boxes(F.MAP_PERSPECTIVE, H, W, 0.0, 0.0, None, coef)
            done += 1
            break
    assert done == 2


@pytest.mark.parametrize("seed", range(8))
def test_certified_fused_boxes_hold_every_tap_also_across_the_inner_clip(seed):
    rng = np.random.default_rng(200 + seed)
    H, W = int(rng.integers(200, 520)), int(rng.integers(300, 900))
    xc, yc = rng.uniform(0.2, 0.8) * W, rng.uniform(0.2, 0.8) * H
    for _ in range(20):
        fact = mild_fact(rng, H, W, int(rng.integers(4, 6)))
        coef = mild_coef(rng, H, W, shift=0.25)            # the frame's image leaves the frame: the inner clip is active on a band of tiles
        lv = level(F.MAP_FUSED, H, W, xc, yc, fact, coef)
        assert lv in (0, 2)
        if lv == 2:
            cut = check_boxes(F.MAP_FUSED, H, W, xc, yc, fact, coef)
            if cut > 0:
                return
    pytest.fail("no certified fused calibration whose inner clip cuts through a tile was drawn")


def test_what_the_certificate_refuses():
    H, W = 400, 700
    strong = [1.0, 0.0, 3e-5]                              # folds: B grows by 3e-5 r^2
    assert level(F.MAP_RADIAL, H, W, 350.0, 200.0, strong, None) == 0
    sign_change = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, -2.0 / W, 0.0]      # the denominator crosses zero inside the frame
    assert level(F.MAP_PERSPECTIVE, H, W, 0.0, 0.0, None, sign_change) == 0
    assert level(F.MAP_FUSED, H, W, 350.0, 200.0, [1.0, 1e-6], sign_change) == 0
    assert level(F.MAP_FUSED, H, W, 350.0, 200.0, strong, [1, 0, 0, 0, 1, 0, 0, 0]) == 0
    # the cfg3 calibration of BASELINE config 3 at full size: certified (what bench.py's cfg3_fused entry relies on)
    from discorpy_amd import configs
    c = configs.cfg3()
    assert level(F.MAP_FUSED, c["shape"][0], c["shape"][1], c["xcenter"], c["ycenter"], c["list_fact"], c["list_coef"]) == 2
