"""MI355X drop-in for the remap functions of ``discorpy.post.postprocessing``.

Same names, positional arguments, defaults and error messages as the reference
(``/root/reference/discorpy/post/postprocessing.py``):

* :func:`unwarp_image_backward`         reference lines 111-148
* :func:`unwarp_slice_backward`         reference lines 188-229
* :func:`unwarp_chunk_slices_backward`  reference lines 255-313
* :func:`correct_perspective_image`     reference lines 462-492

plus :func:`unwarp_perspective_fused`, the one-pass composition BASELINE config 3 asks for.

Every call runs a hand-written HIP kernel through the C ABI of ``libdiscorpy_hip.so``
(``include/discorpy_hip.h``).  There is no CPU path: a missing library or GPU raises.

Inputs may be

* ``numpy.ndarray`` (host): the library stages host<->device copies itself and a NumPy array is
  returned -- the plumbing-compatible mode, dominated by PCIe;
* a ``torch.Tensor`` on a ROCm device: zero-copy, the kernel is enqueued on torch's current
  stream and a tensor on the same device is returned -- the mode the throughput numbers are for;
* any other device array exposing ``__cuda_array_interface__`` (CuPy, Numba): zero-copy on the current
  device; the result is a :class:`discorpy_amd._ffi.DeviceArray` exposing the same interface
  (``cupy.asarray(result)`` wraps it without a copy), or the caller's ``out=``.

What differs from the reference, on purpose:

* float32 data (what ``discorpy.losa.load_image`` / ``load_hdf_file`` produce) takes the tuned kernels;
  float64, (u)int8, (u)int16 and (u)int32 data run the generic kernels with scipy's exact arithmetic and
  its integer rounding, output dtype = input dtype as in the reference -- 64-bit integers (read as doubles, stored as the
  reference's cast stores them on x86-64) and bool included; complex images and chunks go through as their real and imaginary
  parts, as scipy treats them; float16 raises scipy's own ``RuntimeError("data type not supported")``;
* for spline ``order`` 0 and 1 ``mode`` cannot influence the result because every coordinate is
  clipped into the image first; it is validated and otherwise ignored.
  Orders 2..5 run scipy's prefiltered B-spline interpolation on the GPU with all eight modes
  (within one float32 ulp of scipy);
* keyword-only extras: ``out=`` writes into a caller-supplied array / tensor of the right shape and
  dtype; without it NumPy outputs are leased from a recycling pool (``discorpy_amd/_pool.py``) so that
  a loop does not pay 4 ms of page faults per 4096 x 4096 frame; ``blend`` selects the bilinear arithmetic (``"f64lerp"`` default: float64
  factorised lerp, within one float32 ulp of scipy and bit-equal in practice; ``"scipy"``:
  scipy's exact float64 operation order; ``"f32"``: float32 lerp, opt-in).
"""
import ctypes as C
import os

import numpy as np

from .. import _ffi as F
from .. import _pool

__all__ = ["unwarp_line_forward", "unwarp_image_backward", "unwarp_images_backward", "unwarp_slice_backward", "unwarp_chunk_slices_backward",
           "unwarp_slice_backward_centres", "unwarp_chunk_slices_backward_centres",
           "correct_perspective_image", "unwarp_perspective_fused", "remap_coordinates",
           "generate_radial_map", "generate_fused_map"]

_MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")


# --------------------------------------------------------------------------- array plumbing

def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch" and hasattr(a, "data_ptr")


def _is_cai(a):
    """A device array that is not a torch tensor (CuPy, Numba, ...)."""
    return hasattr(a, "__cuda_array_interface__") and not _is_torch(a)


def _cai_dtype(a):
    return np.dtype(a.__cuda_array_interface__["typestr"])


def _is_device_coords(a):
    """A C-contiguous float32 / float64 device array (``__cuda_array_interface__``) usable as a coordinate array in place."""
    if not _is_cai(a):
        return False
    cai = a.__cuda_array_interface__
    dt = np.dtype(cai["typestr"])
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        return False
    st = cai.get("strides")
    if st is None:
        return True
    acc = dt.itemsize
    for n, v in zip(reversed(cai["shape"]), reversed(st)):
        if int(n) > 1 and int(v) != acc:
            return False
        acc *= max(int(n), 1)
    return True


def _cai_is_contiguous(a):
    """C-contiguous ``__cuda_array_interface__`` array of any element type (no strides, or the strides of C order)."""
    cai = a.__cuda_array_interface__
    st = cai.get("strides")
    if st is None:
        return True
    acc = np.dtype(cai["typestr"]).itemsize
    for n, v in zip(reversed(cai["shape"]), reversed(st)):
        if int(n) > 1 and int(v) != acc:
            return False
        acc *= max(int(n), 1)
    return True


def _cai_to_host(a):
    """Host copy of a ``__cuda_array_interface__`` array (C-contiguous ones only; anything else should be copied by its owner)."""
    if hasattr(a, "copy_to_host"):
        return a.copy_to_host()
    if hasattr(a, "get"):
        return a.get()
    cai = a.__cuda_array_interface__
    if not _cai_is_contiguous(a):
        raise ValueError("a strided device coordinate array cannot be read back; pass a contiguous one")
    out = np.empty(tuple(int(v) for v in cai["shape"]), np.dtype(cai["typestr"]))
    F.check(F.lib().dcp_memcpy(out.ctypes.data, int(cai["data"][0]), out.nbytes, F.COPY_D2H, -1, None))
    return out


def _default_blend(host=False):
    """``blend=None``: host (NumPy) arrays get scipy's exact operation order -- the result is then the reference's bit for bit given
    the same float32 coordinates, and the kernel hides under the PCIe transfers anyway; device-resident arrays get the
    factorised float64 form (<= 1 float32 ulp from scipy's, 5-10 % faster where the kernel is what is timed).  The environment
    variable DISCORPY_AMD_BLEND overrides both, DISCORPY_AMD_HOST_BLEND the host default only."""
    return os.environ.get("DISCORPY_AMD_BLEND", os.environ.get("DISCORPY_AMD_HOST_BLEND", "scipy") if host else "f64lerp")


def _blend_code(blend, host=False):
    name = _default_blend(host) if blend is None else blend
    try:
        return F.BLEND_BY_NAME[str(name).lower()]
    except KeyError:
        raise ValueError("unknown blend %r (expected one of %s)" % (name, sorted(F.BLEND_BY_NAME)))


SPLINE_SCIPY_SUM = 0x100      # DCP_SPLINE_SCIPY_SUM: OR-ed into the boundary mode of the spline orders


def _spline_mode(mode, blend):
    """Boundary-mode code of an order >= 2 call; blend="scipy" asks for scipy's tap-by-tap summation order (the default sums the
    taps factorised in the LDS-staged gather: ~1 pixel in 1e8 differs in its last float32 bit)."""
    code = _MODES.index(mode)
    return code | SPLINE_SCIPY_SUM if (blend is not None and str(blend).lower() in ("scipy", "exact")) else code


def _check_order_mode(order, mode):
    if mode not in _MODES:
        # scipy's own message for an unknown boundary mode
        raise RuntimeError("boundary mode not supported")
    import operator
    order = operator.index(order)          # scipy takes an integer: order=1.0 is a TypeError there too
    if order < 0 or order > 5:
        raise RuntimeError("spline order not supported")
    return order


def _complex_parts(a):
    """scipy.ndimage.map_coordinates interpolates a complex array as two real ones (its wrapper calls itself on ``input.real`` and
    ``input.imag`` and fills ``output.real`` / ``output.imag``); the reference hands whatever it is given to it.  Returns
    ``(real, imag, join)`` for a complex NumPy array or torch tensor -- strided views, no copies -- and None for real data."""
    if isinstance(a, np.ndarray):
        if a.dtype.kind != "c":
            return None
        if a.dtype == np.dtype(np.clongdouble) and np.dtype(np.clongdouble) != np.dtype(np.complex128):
            raise RuntimeError("data type not supported")

        def join(re, im):
            res = np.empty(np.shape(re), a.dtype)
            res.real, res.imag = re, im
            return res
        return a.real, a.imag, join
    if _is_torch(a) and a.is_complex():
        import torch
        v = torch.view_as_real(a)
        return v[..., 0], v[..., 1], lambda re, im: torch.complex(re.contiguous(), im.contiguous())
    return None


def _dtype_code(dtype):
    # NumPy: the dtype's name ("uint16" also for a big-endian '>u2' read from HDF5); torch: "torch.uint16"
    name = dtype.name if isinstance(dtype, np.dtype) else str(dtype).replace("torch.", "")
    try:
        return F.DTYPE_BY_NAME[name]
    except KeyError:
        if name in ("float16", "longdouble", "float128", "object", "bfloat16") or name.startswith(("str", "bytes", "void", "datetime", "timedelta")):
            raise RuntimeError("data type not supported")     # scipy's own refusal (the reference passes the array to it)
        raise NotImplementedError("element type %s is not implemented on the GPU path (supported: %s)"
                                  % (dtype, ", ".join(sorted(F.DTYPE_BY_NAME))))


class _Image:
    """A 2-D / 3-D array handed to the C ABI: pointer, element type, element strides, memory kind."""

    def __init__(self, a, ndim):
        self.torch = _is_torch(a)
        self.cai = False
        if self.torch:
            if not a.is_cuda:
                a = a.detach().cpu().numpy()
                self.torch = False
        if not self.torch and _is_cai(a):
            # __cuda_array_interface__ v2/v3: pointer on the current device, byte strides (None = C order), optional stream
            cai = a.__cuda_array_interface__
            dt = np.dtype(cai["typestr"])
            self.cai = True
            self.code = _dtype_code(dt)
            self.dtype = dt
            self.shape = tuple(int(v) for v in cai["shape"])
            st = cai.get("strides")
            if st is None:
                st, acc = [], dt.itemsize
                for n in reversed(self.shape):
                    st.insert(0, acc)
                    acc *= max(int(n), 1)
            if any(v % dt.itemsize or v < 0 for v in st):
                raise ValueError("device array strides must be non-negative multiples of the item size")
            self.strides = tuple(int(v) // dt.itemsize for v in st)
            self.ptr = int(cai["data"][0])
            self.mem = F.MEM_DEVICE
            self.device = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1"))
            stream = cai.get("stream")
            self.stream = None if stream in (None, 1, 2) else int(stream)
            self.keep = a
        elif self.torch:
            import torch
            self.code = _dtype_code(a.dtype)
            self.dtype = a.dtype
            self.shape = tuple(a.shape)
            self.strides = tuple(a.stride())
            self.ptr = a.data_ptr()
            self.mem = F.MEM_DEVICE
            self.device = a.device.index if a.device.index is not None else torch.cuda.current_device()
            self.stream = torch.cuda.current_stream(self.device).cuda_stream
            self.keep = a
        else:
            a = np.asarray(a)
            self.code = _dtype_code(a.dtype)
            if not a.dtype.isnative:
                a = a.astype(a.dtype.newbyteorder("="))
            self.dtype = a.dtype          # native byte order, as scipy allocates its output
            if any(s < 0 for s in a.strides) or any(s % a.itemsize for s in a.strides):
                a = np.ascontiguousarray(a)
            self.shape = a.shape
            self.strides = tuple(s // a.itemsize for s in a.strides)
            self.ptr = a.ctypes.data
            self.mem = F.MEM_HOST
            self.device = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1"))
            self.stream = None
            self.keep = a
        if len(self.shape) != ndim:
            raise ValueError("expected a %d-D array" % ndim)

    @property
    def f32(self):
        return self.code == F.DTYPE_F32

    def empty(self, shape, float32=False, out=None):
        """Fresh output of the same kind (device tensor / NumPy array) and element type as the input
        (float32 if asked) -- or the caller's ``out`` after checking that it is exactly that."""
        if self.torch:
            import torch
            dt = torch.float32 if float32 else self.dtype
            if out is not None:
                if not (_is_torch(out) and out.is_cuda and out.device == self.keep.device and out.dtype == dt
                        and tuple(out.shape) == tuple(shape) and out.is_contiguous()):
                    raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (dt, tuple(shape), self.keep.device))
                return out, out.data_ptr()
            out = torch.empty(shape, dtype=dt, device=self.keep.device)
            return out, out.data_ptr()
        dt = np.dtype(np.float32) if float32 else np.dtype(self.dtype)
        if self.cai:
            if out is not None:
                cai = getattr(out, "__cuda_array_interface__", None)
                if not (cai and np.dtype(cai["typestr"]) == dt and tuple(cai["shape"]) == tuple(shape)
                        and cai.get("strides") is None and not cai["data"][1]):
                    raise ValueError("out must be a writeable C-contiguous %s device array of shape %s" % (dt, tuple(shape)))
                return out, int(cai["data"][0])
            out = F.DeviceArray(shape, dt, self.device)
            return out, out.ptr
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype == dt and out.shape == tuple(shape)
                    and out.flags.c_contiguous and out.flags.writeable):
                raise ValueError("out must be a writeable C-contiguous %s array of shape %s" % (dt, tuple(shape)))
            return out, out.ctypes.data
        out = _pool.empty(shape, dt)      # recycled once the caller drops it (discorpy_amd/_pool.py)
        return out, out.ctypes.data

    def dense_rows(self):
        """2-D image with unit or constant column stride and non-overlapping rows, else a copy."""
        h, w = self.shape
        rs, cs = self.strides
        ok = cs >= 1 and rs >= 1 and (h == 1 or rs >= (w - 1) * cs + 1)
        if ok:
            return self
        if self.cai:
            raise ValueError("device arrays must have positive, non-overlapping strides")
        if self.torch:
            return _Image(self.keep.contiguous(), 2)
        return _Image(np.ascontiguousarray(self.keep), 2)


def _coefs(values, what):
    try:
        vals = [float(v) for v in values]
    except TypeError:
        raise TypeError("%s must be a sequence of numbers" % what)
    return vals


# --------------------------------------------------------------------------- public functions

def unwarp_line_forward(list_lines, xcenter, ycenter, list_fact):
    """
    Unwarp lines of dot-centroids using a forward model (reference ``postprocessing.py:36-64``).

    Parameters
    ----------
    list_lines : list of 2D arrays
        (y, x) coordinates of the dot-centroids of each line.
    xcenter, ycenter : float
        Center of distortion.
    list_fact : list of floats
        Polynomial coefficients of the forward model.

    Returns
    -------
    list of 2D arrays
        The unwarped (y, x) coordinates, line by line.  All points of all lines go through one launch
        (``dcp_map_points_f64``).
    """
    lines = [np.asarray(line) for line in list_lines]
    fa, nf = F.fact_array(_coefs(list_fact, "list_fact"))
    sizes = [len(line) for line in lines]
    if sum(sizes) == 0:
        return [np.zeros_like(line) for line in lines]
    pts = np.ascontiguousarray(np.concatenate([line[:, :2].reshape(-1, 2) for line in lines if len(line)]), dtype=np.float64)
    out = np.empty_like(pts)
    F.require_device()
    F.check(F.lib().dcp_map_points_f64(pts.ctypes.data, out.ctypes.data, pts.shape[0], float(xcenter), float(ycenter), fa, nf,
                                       F.MEM_HOST, int(os.environ.get("DISCORPY_AMD_DEVICE", "-1")), None))
    res, pos = [], 0
    for line, n in zip(lines, sizes):
        uline = np.zeros_like(line)                  # the reference keeps the input's dtype (np.zeros_like)
        if n:
            uline[:, 0] = out[pos:pos + n, 0]
            uline[:, 1] = out[pos:pos + n, 1]
        pos += n
        res.append(uline)
    return res


def correct_perspective_line(list_lines, list_coef):
    """
    Apply perspective correction to lines (reference ``postprocessing.py:414-441``).

    Parameters
    ----------
    list_lines : list of 2D-arrays
        List of the (y,x)-coordinates of points on each line.
    list_coef : list of floats
        Coefficients of the forward-mapping matrix.

    Returns
    -------
    list_clines : list of 2D arrays
        List of the corrected (y,x)-coordinates of points on each line (float64, as the reference builds them).  All points of
        all lines go through one launch (``dcp_map_points_perspective_f64``); bit-equal to the reference.
    """
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    ca, _ = F.fact_array(_coefs(list_coef, "list_coef"))
    lines = [np.asarray(iline) for iline in list_lines]
    sizes = [len(line) for line in lines]
    res = []
    if sum(sizes) == 0:
        return [np.asarray([]) for _ in lines]                 # np.asarray(list(zip(yn, xn))) of an empty line
    pts = np.ascontiguousarray(np.concatenate([line[:, :2].reshape(-1, 2) for line in lines if len(line)]), dtype=np.float64)
    out = np.empty_like(pts)
    F.require_device()
    F.check(F.lib().dcp_map_points_perspective_f64(pts.ctypes.data, out.ctypes.data, pts.shape[0], ca, F.MEM_HOST,
                                                   int(os.environ.get("DISCORPY_AMD_DEVICE", "-1")), None))
    pos = 0
    for n in sizes:
        res.append(out[pos:pos + n].copy() if n else np.asarray([]))
        pos += n
    return res


def unwarp_image_backward(mat, xcenter, ycenter, list_fact, order=1, mode="reflect", *, blend=None, out=None):
    """
    Unwarp an image using a backward model (reference ``postprocessing.py:111-148``).

    Parameters
    ----------
    mat : array_like
        2D array (NumPy array or ROCm torch tensor).
    xcenter : float
        Center of distortion in x-direction.
    ycenter : float
        Center of distortion in y-direction.
    list_fact : list of float
        Polynomial coefficients of the backward model.
    order : int, optional.
        The order of the spline interpolation (0..5).
    mode : {'reflect', 'grid-mirror', 'constant', 'grid-constant', 'nearest',
           'mirror', 'grid-wrap', 'wrap'}, optional
        Boundary mode of scipy's spline interpolation; inert for order <= 1.

    Returns
    -------
    array_like
        2D array. Distortion-corrected image, same kind and dtype as the input.
    """
    (height, width) = mat.shape
    parts = _complex_parts(mat)
    if parts is not None:
        res = parts[2](unwarp_image_backward(parts[0], xcenter, ycenter, list_fact, order, mode, blend=blend),
                       unwarp_image_backward(parts[1], xcenter, ycenter, list_fact, order, mode, blend=blend))
        if out is not None:
            out[...] = res
            return out
        return res
    order = _check_order_mode(order, mode)
    img = _Image(mat, 2).dense_rows()
    bcode = _blend_code(blend, img.mem == F.MEM_HOST)
    fact = _coefs(list_fact, "list_fact")
    fa, nf = F.fact_array(fact)
    out, optr = img.empty((height, width), out=out)
    F.require_device()
    if not img.f32:
        F.check(F.lib().dcp_unwarp_image_typed(img.ptr, optr, img.code, height, width, img.strides[0], img.strides[1],
                                               float(xcenter), float(ycenter), fa, nf, order, _spline_mode(mode, blend),
                                               img.mem, img.device, img.stream))
        return out
    if order >= 2:
        F.check(F.lib().dcp_unwarp_image_spline_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1],
                                                    float(xcenter), float(ycenter), fa, nf, order, _spline_mode(mode, blend),
                                                    img.mem, img.device, img.stream))
        return out
    F.check(F.lib().dcp_unwarp_image_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1],
                                         float(xcenter), float(ycenter), fa, nf, order, 1, bcode,
                                         img.mem, img.device, img.stream))
    return out


def _reshaped_out(out, shape):
    """`out` viewed with `shape` -- or a ValueError: reshaping a non-contiguous array (or tensor) gives a COPY, and the result
    would silently land in it instead of the caller's array."""
    if isinstance(out, np.ndarray):
        if not out.flags.c_contiguous:
            raise ValueError("out must be C-contiguous")
    elif _is_torch(out):
        if not out.is_contiguous():
            raise ValueError("out must be contiguous")
    return out.reshape(shape)


def _per_frame(value, n, what):
    """`value` as n floats: a number is shared by all frames, a sequence must hold one number per frame."""
    if np.ndim(value) == 0:
        return [float(value)] * n
    vals = [float(v) for v in value]
    if len(vals) != n:
        raise ValueError("%s: expected one value or %d (one per image), got %d" % (what, n, len(vals)))
    return vals


def unwarp_images_backward(mats, xcenter, ycenter, list_fact, order=1, mode="reflect", *, blend=None, out=None):
    """
    :func:`unwarp_image_backward` (reference ``postprocessing.py:111-148``) over a batch of images of one shape in ONE
    call -- the loops the reference's users write around it: per colour channel
    (``examples/readthedocs_demo/demo_06.py:111-113``, ``demo_07.py:58-60``), per camera, per candidate calibration.

    Parameters
    ----------
    mats : sequence of 2D arrays, or one 3D array (n, height, width)
        The images (NumPy arrays or ROCm torch tensors), all of one shape.
    xcenter, ycenter : float or sequence of n floats
        Center of distortion, shared or one per image.
    list_fact : list of float, or sequence of n such lists
        Polynomial coefficients of the backward model, shared or one vector per image (lengths may differ).
    order, mode : as :func:`unwarp_image_backward`.

    Returns
    -------
    list of 2D arrays (a sequence was given) or one 3D array (a 3D array was given); every image is bit-identical to what
    :func:`unwarp_image_backward` returns for it.  Device-resident float32 images at order 0 / 1 whose calibrations all
    hold the tile certificate go through ONE kernel launch per 55 images (``dcp_unwarp_images_f32``: the drain of one
    frame overlaps the ramp of the next); anything else is processed image by image.
    """
    stacked = hasattr(mats, "shape") and len(mats.shape) == 3
    if hasattr(mats, "shape") and len(mats.shape) != 3:
        raise ValueError("expected a sequence of 2-D images or one 3-D array (n, height, width)")
    frames = [mats[i] for i in range(mats.shape[0])] if stacked else list(mats)
    n = len(frames)
    order = _check_order_mode(order, mode)
    xcs, ycs = _per_frame(xcenter, n, "xcenter"), _per_frame(ycenter, n, "ycenter")
    per_frame_fact = len(list_fact) > 0 and np.ndim(list_fact[0]) > 0
    if per_frame_fact and len(list_fact) != n:
        raise ValueError("list_fact: expected one coefficient vector or %d (one per image), got %d" % (n, len(list_fact)))
    facts = [_coefs(list_fact[i] if per_frame_fact else list_fact, "list_fact") for i in range(n)]
    outs = None
    if out is not None:
        outs = [out[i] for i in range(n)] if (hasattr(out, "shape") and len(out.shape) == 3) else list(out)
        if len(outs) != n:
            raise ValueError("out must hold one array per image")
    if n == 0:
        return mats if stacked else []
    # A (n, height, width) DEVICE array of an integer type or float64 under ONE calibration, bilinear: the frames are the projections
    # of a stack whose every row is wanted -- the stack kernel on that element type (uint16: 0.59 of the HBM peak against 0.27
    # frame by frame), the same pixels (scipy's blend and store either way).  coord_round_f32 = 2 asks for unwarp_image_backward's
    # semantics -- coordinates clipped to the whole image, no row band -- so folding models give what the frame-by-frame calls give.
    # Arrays the stack entry point cannot address in place (column-strided or overlapping views) go frame by frame below.
    # float32 takes the same route inside the C ABI.
    if (stacked and order == 1 and n >= 2 and out is None and not per_frame_fact and np.ndim(xcenter) == 0 and np.ndim(ycenter) == 0
            and (_is_torch(mats) and mats.is_cuda or _is_cai(mats))
            and str(mats.dtype).replace("torch.", "") in ("uint8", "int8", "uint16", "int16", "uint32", "int32", "float64")
            and 2 <= mats.shape[1] <= 65535 and mats.shape[2] >= 2):
        ps, rs, cs = _Image(mats, 3).strides
        if cs == 1 and rs >= mats.shape[2] and ps >= (mats.shape[1] - 1) * rs + mats.shape[2]:
            return _stack_rows(mats, xcs[0], ycs[0], facts[0], 0.0, int(mats.shape[1]), 2, blend)
    imgs = [_Image(f, 2).dense_rows() for f in frames]
    first = imgs[0]
    bcode = _blend_code(blend, first.mem == F.MEM_HOST)
    uniform = all(im.f32 and im.shape == first.shape and im.strides == first.strides and im.mem == first.mem and
                  im.device == first.device and im.stream == first.stream and im.torch == first.torch for im in imgs)
    nf = max(len(f) for f in facts)
    if not (uniform and order <= 1 and nf <= F.MAX_FACT):
        res = [unwarp_image_backward(frames[i], xcs[i], ycs[i], facts[i], order=order, mode=mode, blend=blend,
                                     out=None if outs is None else outs[i]) for i in range(n)]
    else:
        (height, width) = first.shape
        res, optrs = [], []
        if stacked and outs is None and first.torch:
            import torch
            whole = torch.empty((n, height, width), dtype=torch.float32, device=first.keep.device)
            res = [whole[i] for i in range(n)]
            optrs = [whole.data_ptr() + i * height * width * 4 for i in range(n)]
        elif stacked and outs is None and not first.torch and first.mem == F.MEM_DEVICE:
            # a 3-D device array that is not a tensor (CuPy, Numba, DeviceArray): one 3-D DeviceArray, as the docstring promises
            whole = F.DeviceArray((n, height, width), np.float32, first.device)
            res = [whole.frame(i) for i in range(n)]
            optrs = [r.ptr for r in res]
        else:
            whole = None
            for i, im in enumerate(imgs):
                o, optr = im.empty((height, width), out=None if outs is None else outs[i])
                res.append(o)
                optrs.append(optr)
        table = np.zeros((n, max(nf, 1)), np.float64)             # shorter vectors padded with zeros (the result does not change)
        for i, f in enumerate(facts):
            table[i, :len(f)] = f
        sp = (C.c_void_p * n)(*[im.ptr for im in imgs])
        dp = (C.c_void_p * n)(*optrs)
        xa, ya = (C.c_double * n)(*xcs), (C.c_double * n)(*ycs)
        F.require_device()
        F.check(F.lib().dcp_unwarp_images_f32(sp, dp, n, height, width, first.strides[0], first.strides[1], xa, ya,
                                              table.ctypes.data_as(C.POINTER(C.c_double)), nf, order, 1, bcode, first.mem,
                                              first.device, first.stream))
        if whole is not None:
            return whole
    if not stacked:
        return res
    if out is not None and hasattr(out, "shape") and len(out.shape) == 3:
        return out
    if _is_torch(res[0]):
        import torch
        return torch.stack(res)
    if isinstance(res[0], F.DeviceArray):
        return res
    return np.stack(res)


def unwarp_slice_backward(mat3D, xcenter, ycenter, list_fact, index, *, blend=None, devices=None, out=None):
    """
    Generate an unwarped slice [:,index.:] of a 3D dataset, i.e. one unwarped sinogram of a 3D
    tomographic data (reference ``postprocessing.py:188-229``).  Coordinates stay float64, the
    result is float32 of shape (depth, width).  ``devices=[0, 1, ...]`` shards a host (NumPy) float32
    stack along depth over those GPUs of this process.
    """
    if len(mat3D.shape) < 3:
        raise ValueError("Input must be a 3D data")
    (depth, height, width) = mat3D.shape
    if out is not None:
        if tuple(out.shape) != (depth, width):
            raise ValueError("out must have shape (depth, width)")
        out = _reshaped_out(out, (depth, 1, width))
    index = index - 0.0                  # (the reference computes index - ycenter: a str or None is a TypeError)
    res = _stack_rows(mat3D, xcenter, ycenter, list_fact, float(index), 1, False, blend, out_float32=True,
                      devices=devices, out=out)
    return res.reshape((depth, width)) if isinstance(res, F.DeviceArray) else res[:, 0, :]


def unwarp_chunk_slices_backward(mat3D, xcenter, ycenter, list_fact, start_index, stop_index, *, blend=None,
                                 devices=None, out=None):
    """
    Generate a chunk of unwarped slices [:,start_index: stop_index, :] used for tomographic data
    (reference ``postprocessing.py:255-313``).  Rows ``start_index .. stop_index`` INCLUSIVE;
    coordinates are rounded to float32 as the reference does; ``stop_index=-1`` raises, as it
    does in the reference (:285-288).  ``devices=[0, 1, ...]`` shards a host (NumPy) float32 stack along
    depth over those GPUs of this process (projections are independent, :310-312).
    """
    if (len(mat3D.shape) < 3):
        raise ValueError("Input must be a 3D data")
    (depth, height, width) = mat3D.shape
    index_list = np.arange(height, dtype=np.int16)
    if stop_index == -1:
        stop_index = height
    if (start_index not in index_list) or (stop_index not in index_list):
        raise ValueError("Selected index is out of the range")
    nrows = int(stop_index) - int(start_index) + 1
    if nrows < 1:
        # np.arange(start, stop + 1) is empty in the reference: scipy maps zero coordinates and the result is an empty
        # (depth, 0, width) array of the input's type
        if out is not None:
            if tuple(out.shape) != (depth, 0, width):
                raise ValueError("out must have shape %s" % ((depth, 0, width),))
            return out
        if hasattr(mat3D, "new_empty"):                       # torch tensor
            return mat3D.new_empty((depth, 0, width))
        return np.empty((depth, 0, width), dtype=np.dtype(mat3D.dtype))
    parts = _complex_parts(mat3D)
    if parts is not None:         # (dense copies of the two parts: the stack kernels want unit column stride)
        dense = (lambda t: t.contiguous()) if _is_torch(mat3D) else np.ascontiguousarray
        res = parts[2](_stack_rows(dense(parts[0]), xcenter, ycenter, list_fact, float(start_index), nrows, True, blend, devices=devices),
                       _stack_rows(dense(parts[1]), xcenter, ycenter, list_fact, float(start_index), nrows, True, blend, devices=devices))
        if out is not None:
            out[...] = res
            return out
        return res
    return _stack_rows(mat3D, xcenter, ycenter, list_fact, float(start_index), nrows, True, blend, devices=devices,
                       out=out)


def _stack_rows_centres(mat3D, xcenters, ycenters, list_fact, row_start, nrows, round_f32, blend, out_float32, out):
    """Rows of a stack under K candidate centres: ``(K, depth, nrows, width)``.  float32 stacks in memory (NumPy, torch, device
    arrays) go through ONE call of ``dcp_unwarp_stack_rows_centres_f32``; other element types, out-of-core datasets: centre by centre."""
    xcs = [float(v) for v in xcenters]
    ycs = [float(v) for v in ycenters]
    if len(xcs) != len(ycs):
        raise ValueError("xcenters and ycenters must have the same length")
    k = len(xcs)
    lazy = _is_lazy_stack(mat3D)
    vol = None if lazy else _Image(mat3D, 3)
    if not lazy and vol.f32 and k > 0:
        depth, height, width = vol.shape
        ps, rs, cs = vol.strides
        if cs != 1 or rs < width or (depth > 1 and ps < (height - 1) * rs + width):
            if vol.cai:
                raise ValueError("a device stack must have unit column stride and non-overlapping rows / projections")
            vol = _Image(vol.keep.contiguous() if vol.torch else np.ascontiguousarray(vol.keep), 3)
            ps, rs, cs = vol.strides
        fa, nf = F.fact_array(_coefs(list_fact, "list_fact"))
        res, optr = vol.empty((k, depth, nrows, width), True, out=out)
        if depth > 0:
            F.require_device()
            xa, ya = (C.c_double * k)(*xcs), (C.c_double * k)(*ycs)
            F.check(F.lib().dcp_unwarp_stack_rows_centres_f32(vol.ptr, optr, depth, height, width, ps if depth > 1 else height * rs, rs, xa, ya, k,
                                                              fa, nf, float(row_start), int(nrows), int(round_f32), _blend_code(blend, vol.mem == F.MEM_HOST), vol.mem,
                                                              vol.device, vol.stream))
        return res
    blocks = [_stack_rows(mat3D, xcs[i], ycs[i], list_fact, row_start, nrows, round_f32, blend, out_float32=out_float32) for i in range(k)]
    if k and _is_torch(blocks[0]):
        import torch
        res = torch.stack(blocks)
    else:
        res = np.stack(blocks) if k else np.empty((0,) + tuple(mat3D.shape[:1]) + (nrows, mat3D.shape[2]), np.float32 if out_float32 else np.dtype(mat3D.dtype))
    if out is not None:
        out[...] = res
        return out
    return res


def unwarp_slice_backward_centres(mat3D, xcenters, ycenters, list_fact, index, *, blend=None, out=None):
    """
    :func:`unwarp_slice_backward` (reference ``postprocessing.py:188-229``) for K candidate centres of distortion in ONE call:
    the grid search of ``examples/example_05.py:62-65``, which calls the reference 11 x 11 = 121 times on the same
    600-projection stack.  ``xcenters[k], ycenters[k]``: the candidates; returns float32 ``(K, depth, width)``, block ``k``
    bit-identical to ``unwarp_slice_backward(mat3D, xcenters[k], ycenters[k], list_fact, index)``.  A float32 stack is read
    once for all centres (device-resident: one kernel launch per 224 centres; NumPy: one upload of the union of the row bands).
    """
    if len(mat3D.shape) < 3:
        raise ValueError("Input must be a 3D data")
    (depth, height, width) = mat3D.shape
    k = len(xcenters)
    if out is not None:
        if tuple(out.shape) != (k, depth, width):
            raise ValueError("out must have shape (centres, depth, width)")
        out = _reshaped_out(out, (k, depth, 1, width))
    index = index - 0.0
    res = _stack_rows_centres(mat3D, xcenters, ycenters, list_fact, float(index), 1, False, blend, True, out)
    return res.reshape((k, depth, width)) if isinstance(res, F.DeviceArray) else res[:, :, 0, :]


def unwarp_chunk_slices_backward_centres(mat3D, xcenters, ycenters, list_fact, start_index, stop_index, *, blend=None, out=None):
    """
    :func:`unwarp_chunk_slices_backward` (reference ``postprocessing.py:255-313``) for K candidate centres in one call: rows
    ``start_index .. stop_index`` inclusive, float32 coordinates as the reference; returns ``(K, depth, rows, width)``.
    """
    if (len(mat3D.shape) < 3):
        raise ValueError("Input must be a 3D data")
    (depth, height, width) = mat3D.shape
    index_list = np.arange(height, dtype=np.int16)
    if stop_index == -1:
        stop_index = height
    if (start_index not in index_list) or (stop_index not in index_list):
        raise ValueError("Selected index is out of the range")
    nrows = int(stop_index) - int(start_index) + 1
    if nrows < 1:
        # as unwarp_chunk_slices_backward (and the reference's np.arange(start, stop + 1) of nothing): an empty chunk per centre
        k = len(xcenters)
        if out is not None:
            if tuple(out.shape) != (k, depth, 0, width):
                raise ValueError("out must have shape %s" % ((k, depth, 0, width),))
            return out
        if hasattr(mat3D, "new_empty"):
            return mat3D.new_empty((k, depth, 0, width))
        return np.empty((k, depth, 0, width), dtype=np.dtype(mat3D.dtype))
    return _stack_rows_centres(mat3D, xcenters, ycenters, list_fact, float(start_index), nrows, True, blend, False, out)


def _is_lazy_stack(a):
    """An h5py-style dataset: has shape / dtype / slicing but is neither a NumPy array nor a tensor."""
    return (not isinstance(a, np.ndarray) and not _is_torch(a) and not _is_cai(a) and hasattr(a, "shape") and hasattr(a, "dtype")
            and hasattr(a, "__getitem__") and not isinstance(a, (list, tuple)))


def _stack_rows_lazy(src, xcenter, ycenter, list_fact, row_start, nrows, round_f32, blend, out_float32, out):
    """Out-of-core stack (e.g. the h5py dataset ``losa.load_hdf_object`` returns): like the reference
    (``:221-228``, ``:295-301``) only the row band the requested rows can reach is read from each projection --
    ``src[d0:d1, band, :]`` per depth chunk, the next chunk being read while this one is on the GPU."""
    from concurrent.futures import ThreadPoolExecutor
    bcode = _blend_code(blend, True)                  # an out-of-core stack is host data
    (depth, height, width) = src.shape
    dtype = np.dtype(src.dtype)
    code = _dtype_code(dtype)
    dtype = dtype.newbyteorder("=")                   # chunks are converted to native byte order on the way in
    fact = _coefs(list_fact, "list_fact")
    fa, nf = F.fact_array(fact)
    odt = np.dtype(np.float32) if out_float32 else dtype
    if out is None:
        out = _pool.empty((depth, nrows, width), odt)
    elif not (isinstance(out, np.ndarray) and out.dtype == odt and out.shape == (depth, nrows, width)
              and out.flags.c_contiguous and out.flags.writeable):
        raise ValueError("out must be a writeable C-contiguous %s array of shape %s" % (odt, (depth, nrows, width)))
    if depth == 0:
        return out
    F.require_device()
    b0, bn = F.stack_row_band(height, width, xcenter, ycenter, fact, row_start, nrows)
    per = max(bn * width * dtype.itemsize, 1)
    chunk = int(max(1, min(depth, int(float(os.environ.get("DISCORPY_AMD_READ_CHUNK_MB", "64")) * (1 << 20)) // per)))
    device = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1"))

    def read(d0):
        a = np.asarray(src[d0:min(depth, d0 + chunk), b0:b0 + bn, :])
        return np.ascontiguousarray(a, dtype=dtype)

    with ThreadPoolExecutor(max_workers=1) as reader:
        nxt = reader.submit(read, 0)
        for d0 in range(0, depth, chunk):
            band = nxt.result()
            if d0 + chunk < depth:
                nxt = reader.submit(read, d0 + chunk)
            n = band.shape[0]
            F.check(F.lib().dcp_unwarp_stack_band(band.ctypes.data, out[d0:d0 + n].ctypes.data, code, int(out_float32), n,
                                                  height, width, b0, bn, bn * width, width, float(xcenter),
                                                  float(ycenter), fa, nf, float(row_start), nrows, int(round_f32), bcode,
                                                  F.MEM_HOST, device, None))
    return out


def _stack_rows(mat3D, xcenter, ycenter, list_fact, row_start, nrows, round_f32, blend, out_float32=False,
                devices=None, out=None):
    if _is_lazy_stack(mat3D):
        if devices is not None:
            raise ValueError("devices= needs a NumPy stack in memory")
        return _stack_rows_lazy(mat3D, xcenter, ycenter, list_fact, row_start, nrows, round_f32, blend, out_float32, out)
    vol = _Image(mat3D, 3)
    bcode = _blend_code(blend, vol.mem == F.MEM_HOST)
    depth, height, width = vol.shape
    if depth == 0:
        return vol.empty((0, nrows, width), out_float32, out=out)[0]
    ps, rs, cs = vol.strides
    if cs != 1 or rs < width or (depth > 1 and ps < (height - 1) * rs + width):
        if vol.cai:
            raise ValueError("a device stack must have unit column stride and non-overlapping rows / projections")
        vol = _Image(vol.keep.contiguous() if vol.torch else np.ascontiguousarray(vol.keep), 3)
        ps, rs, cs = vol.strides
    if devices is not None:
        if vol.torch or vol.cai:
            raise ValueError("devices= shards a host (NumPy) stack; a device array already lives on one GPU "
                             "(see discorpy_amd.stack for one process per GPU)")
        if not vol.f32:
            raise NotImplementedError("devices= is implemented for float32 stacks (this one is %s)" % (vol.dtype,))
    fa, nf = F.fact_array(_coefs(list_fact, "list_fact"))
    out, optr = vol.empty((depth, nrows, width), out_float32, out=out)
    F.require_device()
    if not vol.f32:
        F.check(F.lib().dcp_unwarp_stack_rows_typed(vol.ptr, optr, vol.code, int(out_float32), depth, height, width,
                                                    ps if depth > 1 else height * rs, rs, float(xcenter),
                                                    float(ycenter), fa, nf, float(row_start), nrows, int(round_f32),
                                                    vol.mem, vol.device, vol.stream))
        return out
    if devices is not None:
        devs = [int(d) for d in devices]
        arr = (C.c_int * max(len(devs), 1))(*devs)
        F.check(F.lib().dcp_unwarp_stack_rows_multi_f32(vol.ptr, optr, depth, height, width,
                                                        ps if depth > 1 else height * rs, rs, float(xcenter),
                                                        float(ycenter), fa, nf, float(row_start), nrows, int(round_f32),
                                                        bcode, arr, len(devs)))
        return out
    F.check(F.lib().dcp_unwarp_stack_rows_f32(vol.ptr, optr, depth, height, width, ps if depth > 1 else height * rs,
                                              rs, float(xcenter), float(ycenter), fa, nf, float(row_start), nrows,
                                              int(round_f32), bcode, vol.mem, vol.device, vol.stream))
    return out


def correct_perspective_image(mat, list_coef, order=1, mode="reflect", map_index=None, *, blend=None, out=None):
    """
    Apply perspective correction to an image (reference ``postprocessing.py:462-492``).

    Parameters
    ----------
    mat : array_like
        2D array. Image for correction.
    list_coef : list of floats
        Coefficients of the backward-mapping matrix (c1..c8, (x, y) convention).
    order : int, optional.
        The order of the spline interpolation (0..5).
    mode : str, optional
        Boundary mode of scipy's spline interpolation; inert for order <= 1.
    map_index : array_like
        Indices for mapping, (ycoords, xcoords) with height*width points. Generated if None.

    Returns
    -------
    array_like
        Corrected image.
    """
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    (height, width) = mat.shape
    parts = _complex_parts(mat)
    if parts is not None:
        res = parts[2](correct_perspective_image(parts[0], list_coef, order, mode, map_index, blend=blend),
                       correct_perspective_image(parts[1], list_coef, order, mode, map_index, blend=blend))
        if out is not None:
            out[...] = res
            return out
        return res
    order = _check_order_mode(order, mode)
    img = _Image(mat, 2).dense_rows()
    bcode = _blend_code(blend, img.mem == F.MEM_HOST)
    if map_index is not None:
        ymap, xmap = map_index[0], map_index[1]
        res = remap_coordinates(mat, ymap, xmap, order=order, mode=mode, blend=blend).reshape((height, width))
        if out is not None:
            out[...] = res
            return out
        return res
    ca, _ = F.fact_array(_coefs(list_coef, "list_coef"))
    out, optr = img.empty((height, width), out=out)
    F.require_device()
    if not img.f32:
        F.check(F.lib().dcp_perspective_image_typed(img.ptr, optr, img.code, height, width, img.strides[0],
                                                    img.strides[1], ca, order, _spline_mode(mode, blend), img.mem, img.device,
                                                    img.stream))
        return out
    if order >= 2:
        F.check(F.lib().dcp_perspective_image_spline_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1],
                                                         ca, order, _spline_mode(mode, blend), img.mem, img.device, img.stream))
        return out
    F.check(F.lib().dcp_perspective_image_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1], ca,
                                              order, bcode, img.mem, img.device, img.stream))
    return out


def _coordinate_map(shape, kind, xcenter, ycenter, list_fact, list_coef, like):
    (height, width) = shape
    fa, nf = F.fact_array(_coefs(list_fact, "list_fact"))
    ca, _ = F.fact_array(_coefs(list_coef, "list_coef")) if list_coef is not None else (None, 0)
    if like is not None and _is_torch(like) and like.is_cuda:
        import torch
        ymap = torch.empty((height, width), dtype=torch.float32, device=like.device)
        xmap = torch.empty((height, width), dtype=torch.float32, device=like.device)
        yp, xp, mem = ymap.data_ptr(), xmap.data_ptr(), F.MEM_DEVICE
        dev = like.device.index if like.device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(dev).cuda_stream
    elif like is not None and _is_cai(like):
        # CuPy / Numba image: device-resident maps exposing the same interface
        dev, stream = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1")), None
        ymap = F.DeviceArray((height, width), np.float32, dev)
        xmap = F.DeviceArray((height, width), np.float32, dev)
        yp, xp, mem = ymap.ptr, xmap.ptr, F.MEM_DEVICE
    else:
        ymap = np.empty((height, width), np.float32)
        xmap = np.empty((height, width), np.float32)
        yp, xp, mem = ymap.ctypes.data, xmap.ctypes.data, F.MEM_HOST
        dev, stream = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1")), None
    F.require_device()
    F.check(F.lib().dcp_coordinate_map_f32(yp, xp, height, width, kind, float(xcenter), float(ycenter), fa, nf, ca, mem,
                                           dev, stream))
    return ymap, xmap


def _generate_perspective_map(mat, list_coef):
    """
    Generate mapping indices between images (reference ``postprocessing.py:444-459``): the tuple
    ``(yd, xd)`` of float32 arrays of shape ``(height*width, 1)`` that ``correct_perspective_image``
    accepts as ``map_index``.
    """
    (height, width) = mat.shape
    ymap, xmap = _coordinate_map((height, width), F.MAP_PERSPECTIVE, 0.0, 0.0, [], list_coef, mat)
    return ymap.reshape((-1, 1)), xmap.reshape((-1, 1))


def generate_radial_map(shape, xcenter, ycenter, list_fact, *, like=None):
    """
    The float32 ``(yd_mat, xd_mat)`` planes of ``unwarp_image_backward`` (reference :141-145) for an
    image of ``shape`` -- e.g. for ``cv2.remap(img, xd_mat, yd_mat, ...)`` as
    ``discorpy/util/utility.py:425-435`` does, or for ``remap_coordinates``.  ``like`` = a device
    tensor to get device-resident maps.
    """
    return _coordinate_map(tuple(shape), F.MAP_RADIAL, xcenter, ycenter, list_fact, None, like)


def generate_fused_map(shape, xcenter, ycenter, list_fact, list_coef, *, like=None):
    """The composed perspective -> radial map of :func:`unwarp_perspective_fused`, as (yd, xd)."""
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    return _coordinate_map(tuple(shape), F.MAP_FUSED, xcenter, ycenter, list_fact, list_coef, like)


def unwarp_perspective_fused(mat, xcenter, ycenter, list_fact, list_coef, order=1, mode="reflect", *, blend=None,
                             out=None):
    """
    Perspective and radial correction in ONE resampling (BASELINE config 3).

    For output pixel (x, y): (xp, yp) = float32(clip(H(x, y))) exactly as ``_generate_perspective_map``
    (reference :448-457), then the radial backward map of :141-145 evaluated at (xp, yp),
    float32-rounded, and one sample of ``mat``.  This is NOT equal to
    ``correct_perspective_image(unwarp_image_backward(mat, ...), list_coef)``, which resamples
    twice (reference ``examples/readthedocs_demo/demo_05.py:127,147``); on a noise image the two
    differ by up to 0.5.  Its oracle is one ``map_coordinates`` call at the composed coordinates.
    """
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    (height, width) = mat.shape
    order = _check_order_mode(order, mode)
    img = _Image(mat, 2).dense_rows()
    bcode = _blend_code(blend, img.mem == F.MEM_HOST)
    fa, nf = F.fact_array(_coefs(list_fact, "list_fact"))
    ca, _ = F.fact_array(_coefs(list_coef, "list_coef"))
    out, optr = img.empty((height, width), out=out)
    F.require_device()
    if not img.f32:
        F.check(F.lib().dcp_unwarp_fused_typed(img.ptr, optr, img.code, height, width, img.strides[0], img.strides[1],
                                               float(xcenter), float(ycenter), fa, nf, ca, order, _spline_mode(mode, blend),
                                               img.mem, img.device, img.stream))
        return out
    if order >= 2:
        F.check(F.lib().dcp_unwarp_fused_spline_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1],
                                                    float(xcenter), float(ycenter), fa, nf, ca, order, _spline_mode(mode, blend),
                                                    img.mem, img.device, img.stream))
        return out
    F.check(F.lib().dcp_unwarp_fused_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1],
                                         float(xcenter), float(ycenter), fa, nf, ca, order, bcode,
                                         img.mem, img.device, img.stream))
    return out


def remap_coordinates(mat, ycoords, xcoords, order=1, mode="reflect", *, blend=None):
    """
    ``scipy.ndimage.map_coordinates(mat, (ycoords, xcoords), order, mode)`` for coordinates that lie
    inside the image (the ``map_index=`` path of ``correct_perspective_image`` :489-491 and
    ``_mapping`` :250-251).  Coordinates outside the image are treated as scipy treats them under ``mode`` at every order
    (coordinate mapping, tap folding, ``cval = 0`` for the two constant modes).  Returns an array shaped like ``ycoords``.
    """
    (height, width) = mat.shape
    order = _check_order_mode(order, mode)
    img = _Image(mat, 2).dense_rows()
    bcode = _blend_code(blend, img.mem == F.MEM_HOST)
    staged = None
    if img.torch:
        import torch
        yc = torch.as_tensor(ycoords, device=img.keep.device)
        xc = torch.as_tensor(xcoords, device=img.keep.device)
        dt = torch.float32 if (yc.dtype == torch.float32 and xc.dtype == torch.float32) else torch.float64
        yc = yc.to(dt).contiguous()
        xc = xc.to(dt).contiguous()
        if yc.numel() != xc.numel():
            raise ValueError("the two coordinate arrays differ in size (numpy cannot stack them: inhomogeneous shape)")
        cdt = F.COORD_F32 if dt == torch.float32 else F.COORD_F64
        npts, yptr, xptr, shape = yc.numel(), yc.data_ptr(), xc.data_ptr(), tuple(yc.shape)
    elif img.cai and _is_device_coords(ycoords) and _is_device_coords(xcoords) and \
            _cai_dtype(ycoords) == _cai_dtype(xcoords):
        # device image, device coordinates (CuPy / Numba arrays, or the DeviceArray maps of _generate_perspective_map):
        # contiguous float32 / float64, used in place
        yci, xci = ycoords.__cuda_array_interface__, xcoords.__cuda_array_interface__
        dt = _cai_dtype(ycoords)
        shape = tuple(int(v) for v in yci["shape"])
        npts = int(np.prod(shape, dtype=np.int64))
        if npts != int(np.prod(xci["shape"], dtype=np.int64)):
            raise ValueError("the two coordinate arrays differ in size (numpy cannot stack them: inhomogeneous shape)")
        cdt = F.COORD_F32 if dt == np.float32 else F.COORD_F64
        yptr, xptr = int(yci["data"][0]), int(xci["data"][0])
        staged = (ycoords, xcoords)
    else:
        if _is_cai(ycoords):
            ycoords = _cai_to_host(ycoords)
        if _is_cai(xcoords):
            xcoords = _cai_to_host(xcoords)
        yc, xc = np.asarray(ycoords), np.asarray(xcoords)
        dt = np.float32 if (yc.dtype == np.float32 and xc.dtype == np.float32) else np.float64
        yc = np.ascontiguousarray(yc, dtype=dt)
        xc = np.ascontiguousarray(xc, dtype=dt)
        if yc.size != xc.size:
            raise ValueError("the two coordinate arrays differ in size (numpy cannot stack them: inhomogeneous shape)")
        cdt = F.COORD_F32 if dt == np.float32 else F.COORD_F64
        npts, yptr, xptr, shape = yc.size, yc.ctypes.data, xc.ctypes.data, yc.shape
        if img.mem == F.MEM_DEVICE:
            # device image (a __cuda_array_interface__ array), host coordinates: the kernel reads the coordinates on the
            # device, so they are uploaded first (the buffers live until the call has been enqueued and are freed -- which
            # waits for the device -- when `staged` goes out of scope)
            staged = (F.DeviceBuffer(max(yc.nbytes, 4), img.device).upload(yc), F.DeviceBuffer(max(xc.nbytes, 4), img.device).upload(xc))
            yptr, xptr = staged[0].ptr, staged[1].ptr
    out, optr = img.empty(shape)
    F.require_device()
    if not img.f32:
        F.check(F.lib().dcp_remap_coords_typed(img.ptr, optr, img.code, height, width, img.strides[0], img.strides[1],
                                               yptr, xptr, cdt, npts, order, _spline_mode(mode, blend), img.mem, img.device,
                                               img.stream))
        return out
    if order >= 2:
        F.check(F.lib().dcp_remap_coords_spline_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1], yptr,
                                                    xptr, cdt, npts, order, _spline_mode(mode, blend), img.mem, img.device,
                                                    img.stream))
        return out
    F.check(F.lib().dcp_remap_coords_mode_f32(img.ptr, optr, height, width, img.strides[0], img.strides[1], yptr, xptr,
                                              cdt, npts, order, _MODES.index(mode), bcode, img.mem, img.device, img.stream))
    return out
