"""Colour / padded unwarp on the GPU path.

``unwarp_color_image_backward`` has the signature and semantics of the reference
(``/root/reference/discorpy/util/utility.py:278-342``): optional padding, then the backward radial
unwarp of every channel at the SAME float32 coordinates -- one launch on the interleaved image for
orders 0 / 1, plane by plane through :mod:`discorpy_amd.post.postprocessing` otherwise.

``find_point_to_point`` is the closed-form point mapping of ``utility.py:192-230`` (host, NumPy);
``transform_coef_backward_and_forward`` the small least-squares model reversal ``pad=True`` needs
(``proc/processing.py:615-674``; host, NumPy -- the north star keeps the one-off fits on the CPU).
"""
import numpy as np

from ..post import postprocessing as _pp

__all__ = ["unwarp_color_image_backward", "find_point_to_point", "transform_coef_backward_and_forward"]


def find_point_to_point(points, xcenter, ycenter, list_fact, output_order="xy"):
    """
    The radial model applied to ONE point (reference ``utility.py:192-230``): the point moves along its ray
    from the centre by the factor ``B(r) = sum_i list_fact[i] * r**i``.  With a forward model a distorted
    point goes to the undistorted space, with a backward model the other way.  ``points`` is
    ``(row_index, column_index)``; returns ``(x, y)``, or ``(y, x)`` for ``output_order="yx"``.
    """
    row, col = points[0], points[1]
    dx = col - xcenter
    dy = row - ycenter
    radius = np.sqrt(dx * dx + dy * dy)
    coefs = np.asarray(list_fact, dtype=np.float64)
    scale = np.float64(np.sum(coefs * radius ** np.arange(coefs.size)))
    moved = (xcenter + scale * dx, ycenter + scale * dy)
    return moved if output_order == "xy" else moved[::-1]


def transform_coef_backward_and_forward(list_fact, mapping="backward", ref_points=None):
    """
    Coefficients of the reversed radial model (reference ``proc/processing.py:615-674``, the one-off CPU fit the
    north star leaves on the host).  A model maps a radius ``r`` to ``r * F(r)``, ``F(r) = sum_i list_fact[i] r**i``;
    the reversed model ``G`` satisfies ``G(r F(r)) = 1 / F(r)``, and its coefficients are the least-squares solution
    of that identity over the radii of ``ref_points`` (``(y, x)`` offsets from the centre; a 40 x 40 grid of
    +-1000 px in steps of 50 when not given).  ``mapping`` only names which of the two directions ``list_fact``
    is -- the arithmetic is the same.  Points where ``F`` vanishes are left out.
    """
    coefs = np.asarray(list_fact, dtype=np.float64)
    if ref_points is None:
        ticks = np.arange(-1000, 1000, 50)
        pts = np.stack(np.meshgrid(ticks, ticks, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    else:
        if len(ref_points) < coefs.size:
            raise ValueError("Number of reference-points must be equal or "
                             "larger than the number of coefficients!!!")
        pts = np.asarray(ref_points, dtype=np.float64).reshape(-1, 2)
    if mapping not in ("backward", "forward"):
        raise ValueError("mapping must be 'backward' or 'forward'")
    expo = np.arange(coefs.size, dtype=np.int16)
    radius = np.sqrt(pts[:, 1] * pts[:, 1] + pts[:, 0] * pts[:, 0])
    scale = np.sum(coefs[None, :] * np.power(radius[:, None], expo[None, :]), axis=1)
    keep = scale != 0.0
    moved = scale[keep] * radius[keep]
    design = np.power(moved[:, None], expo[None, :])
    return np.linalg.lstsq(design, 1.0 / scale[keep], rcond=1e-64)[0]


def _auto_pad(height, width, xcenter, ycenter, list_fact):
    """pad=True (reference ``utility.py:238-263``): how far the four image corners land outside the frame once the
    forward model -- fitted from the backward one on a 40 x 40 grid over the frame -- is applied to them."""
    grid = [[gy - ycenter, gx - xcenter] for gy in np.linspace(0, height, 40) for gx in np.linspace(0, width, 40)]
    forward = transform_coef_backward_and_forward(list_fact, ref_points=grid)
    corners = {name: find_point_to_point(rc, xcenter, ycenter, forward)
               for name, rc in (("tl", (0, 0)), ("tr", (0, width - 1)), ("br", (height - 1, width - 1)),
                                ("bl", (height - 1, 0)))}
    left = min(corners["tl"][0], corners["bl"][0])
    right = max(corners["tr"][0], corners["br"][0])
    top = min(corners["tl"][1], corners["tr"][1])
    bottom = max(corners["bl"][1], corners["br"][1])
    return (int(-top) if top < 0 else 0, int(bottom - height) if bottom > height else 0,
            int(-left) if left < 0 else 0, int(right - width) if right > width else 0)


def _calc_pad(pad, height, width, xcenter, ycenter, list_fact):
    """(top, bottom, left, right) pad widths from the ``pad`` argument of the reference (``utility.py:233-275``):
    False -> none, True -> automatic, an int -> that width on every side, four values -> as given."""
    if isinstance(pad, bool):
        return _auto_pad(height, width, xcenter, ycenter, list_fact) if pad else (0, 0, 0, 0)
    if isinstance(pad, int):
        return (pad,) * 4
    if isinstance(pad, (tuple, list)):
        if len(pad) != 4:
            raise ValueError("Incorrect format!!! Please use a tuple/list of "
                             "(top_pad, bottom_pad, left_pad, right_pad)")
        return tuple(pad)
    raise ValueError("Invalid format of the 'pad' parameter!!!")


def _pad_device(t, pad_width, mode):
    """numpy.pad for a ROCm tensor: source indices come from numpy.pad on an index ramp, so the
    'edge', 'reflect', 'symmetric' and 'wrap' modes follow numpy's definition exactly."""
    import torch
    if all(p == (0, 0) for p in pad_width):
        return t
    if mode == "constant":
        out = torch.zeros([s + a + b for s, (a, b) in zip(t.shape, pad_width)], dtype=t.dtype, device=t.device)
        sl = tuple(slice(a, a + s) for s, (a, _) in zip(t.shape, pad_width))
        out[sl] = t
        return out
    if mode not in ("edge", "reflect", "symmetric", "wrap"):
        raise NotImplementedError("pad_mode %r is not implemented for device tensors (use a NumPy array, "
                                  "or one of constant/edge/reflect/symmetric/wrap)" % mode)
    for axis, (a, b) in enumerate(pad_width):
        if a or b:
            idx = np.pad(np.arange(t.shape[axis]), (a, b), mode=mode)
            t = t.index_select(axis, torch.as_tensor(idx, device=t.device))
    return t


def _unwarp_interleaved(mat_pad, xcenter, ycenter, list_fact, order, blend=None):
    """(H, W, C) interleaved image in one launch of dcp_unwarp_color_image: one coordinate per pixel, C blends in the
    arithmetic `blend` names (the default of unwarp_image_backward when None; integer pixels always in scipy's exact order),
    no per-channel planes on the host.  3 / 4 channels of float32 / uint8 / uint16 under a certified calibration run
    remap_wg_color_kernel (the source box of a tile staged in LDS once for all channels)."""
    F = _pp.F
    img = _pp._Image(mat_pad, 3)
    h, w, c = img.shape
    rs, ps, cs = img.strides
    if cs != 1 or ps < c or (h > 1 and rs < (w - 1) * ps + c):
        img = _pp._Image(img.keep.contiguous() if img.torch else np.ascontiguousarray(img.keep), 3)
        rs, ps, cs = img.strides
    fa, nf = F.fact_array(_pp._coefs(list_fact, "list_fact"))
    out, optr = img.empty((h, w, c))
    F.require_device()
    F.check(F.lib().dcp_unwarp_color_image(img.ptr, optr, img.code, h, w, c, rs, ps, float(xcenter), float(ycenter),
                                           fa, nf, order, _pp._blend_code(blend, img.mem == F.MEM_HOST), img.mem, img.device, img.stream))
    return out


def unwarp_color_image_backward(mat, xcenter, ycenter, list_fact, order=1, mode="reflect", pad=False,
                                pad_mode='constant', *, blend=None):
    """
    Unwarp a color image using a backward model (reference ``utility.py:278-342``).

    Parameters
    ----------
    mat : array_like
        2D/3D float32 array (H, W) or (H, W, C); NumPy array or ROCm torch tensor.
    xcenter, ycenter : float
        Center of distortion (of the UNPADDED image, as in the reference).
    list_fact : list of float
        Polynomial coefficients of the backward model.
    order : int, optional.
        The order of the spline interpolation (0 or 1).
    mode : str, optional
        Accepted for compatibility; inert for order <= 1.
    pad : bool, int, or tuple of int.
        Use to keep the original view.  ``True``: the width is calculated automatically.
    pad_mode : str
        numpy.pad mode ('constant', 'reflect', 'edge', 'mean', 'linear_ramp', 'symmetric', ...).

    Returns
    -------
    array_like
        2D/3D array. Distortion-corrected image, shape of the padded input.
    """
    (height, width) = mat.shape[:2]
    t_pad, b_pad, l_pad, r_pad = _calc_pad(pad, height, width, xcenter, ycenter, list_fact)
    num_dim = len(mat.shape)
    if num_dim == 2:
        pad_width = [(t_pad, b_pad), (l_pad, r_pad)]
    else:
        pad_width = [(t_pad, b_pad), (l_pad, r_pad), (0, 0)]
    is_torch = _pp._is_torch(mat) and mat.is_cuda
    if is_torch:
        mat_pad = _pad_device(mat, pad_width, pad_mode)
    elif t_pad == b_pad == l_pad == r_pad == 0:
        mat_pad = np.asarray(mat)                     # np.pad would copy the image for nothing
    else:
        mat_pad = np.pad(np.asarray(mat), pad_width, mode=pad_mode)
    xcenter = xcenter + l_pad
    ycenter = ycenter + t_pad
    if num_dim == 2:
        return _pp.unwarp_image_backward(mat_pad, xcenter, ycenter, list_fact, order=order, mode=mode, blend=blend)
    order = _pp._check_order_mode(order, mode)
    parts = _pp._complex_parts(mat_pad)
    if parts is not None:         # scipy interpolates a complex array as its real and imaginary parts
        dense = (lambda t: t.contiguous()) if is_torch else np.ascontiguousarray
        return parts[2](unwarp_color_image_backward(dense(parts[0]), xcenter, ycenter, list_fact, order, mode, blend=blend),
                        unwarp_color_image_backward(dense(parts[1]), xcenter, ycenter, list_fact, order, mode, blend=blend))
    # (`blend` names are case-insensitive everywhere, here too: "SciPy" must not fall to the plane-by-plane path -- ADVICE r4)
    if order <= 1 and (blend is None or str(blend).lower() in ("scipy", "exact", "f64lerp", "f64")) and 1 <= mat_pad.shape[2] <= 64:
        return _unwarp_interleaved(mat_pad, xcenter, ycenter, list_fact, order, blend)
    # channels as dense planes through the batched entry point (the reference's loop over mat_pad[:, :, i], utility.py:320-341):
    # device-resident float32 planes at order 0 / 1 share ONE launch, the other cases go plane by plane inside it
    if is_torch:
        planes = mat_pad.permute(2, 0, 1).contiguous()
        out = _pp.unwarp_images_backward(planes, xcenter, ycenter, list_fact, order=order, mode=mode, blend=blend)
        return out.permute(1, 2, 0)
    planes = np.ascontiguousarray(np.moveaxis(mat_pad, 2, 0))
    mat_corr = _pp.unwarp_images_backward(planes, xcenter, ycenter, list_fact, order=order, mode=mode, blend=blend)
    return np.moveaxis(np.asarray(mat_corr), 0, 2)
