"""ctypes binding of oracle/libunwarp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of oracle/unwarp_oracle.c).  The functions mirror
the reference signatures of discorpy/post/postprocessing.py (:111, :188, :255,
:462); float32 data has dedicated entry points, the other element types go through
map_coordinates().
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB_PATH: another build of the same source, e.g. the AddressSanitizer / UBSan one (`make -C oracle asan`, tools/oracle_asan.sh)
_LIB_PATH = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "libunwarp_oracle.so")

POLY_KERNEL, POLY_NUMPY, POLY_KERNEL_MULADD = 0, 1, 2
BLEND_SCIPY, BLEND_F64LERP, BLEND_F32LERP = 0, 1, 2


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "unwarp_oracle.c")
    if os.environ.get("ORACLE_LIB_PATH"):
        return _LIB_PATH
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.run(["make", "-C", _HERE, "-B", "libunwarp_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i64, dbl, i32 = C.c_int64, C.c_double, C.c_int
        fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
        L.orc_set_threads.argtypes = [i32]
        L.orc_max_threads.restype = i32
        L.orc_get_threads.restype = i32
        L.orc_radial_coords.argtypes = [i64, i64, dbl, dbl, dp, i32, i32, i32, dp, dp]
        L.orc_radial_coords_rows.argtypes = [i64, i64, dbl, dbl, dp, i32, i32, i32, dbl, i64, dp, dp]
        L.orc_perspective_coords.argtypes = [i64, i64, dp, i32, dp, dp]
        L.orc_unwarp_image_f32.argtypes = [fp, fp, i64, i64, i64, dbl, dbl, dp, i32, i32, i32, i32, i32]
        L.orc_perspective_image_f32.argtypes = [fp, fp, i64, i64, i64, dp, i32, i32]
        L.orc_unwarp_fused_f32.argtypes = [fp, fp, i64, i64, i64, dbl, dbl, dp, i32, dp, i32, i32, i32]
        L.orc_remap_coords_f32.argtypes = [fp, fp, i64, i64, i64, vp, vp, i32, i64, i32, i32, i32]
        L.orc_chunk_band.argtypes = [i64, i64, dbl, dbl, dp, i32, dbl, dbl, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_chunk_band.restype = None
        L.orc_unwarp_stack_rows_f32.argtypes = [fp, fp, i64, i64, i64, dbl, dbl, dp, i32, dbl, i64,
                                                i32, i32, i32]
        L.orc_spline_pad.argtypes = [i32]
        L.orc_spline_coefficients_f32.argtypes = [fp, i64, i64, i64, i32, i32, dp]
        L.orc_remap_spline_f32.argtypes = [fp, fp, i64, i64, i64, i32, dbl, dbl, dp, i32, dp, vp, vp, i32, i64, i32, i32,
                                           i32, dp]
        L.orc_map_coordinates_typed.argtypes = [vp, vp, i32, i64, i64, i64, vp, vp, i32, i64, i32, i32, dp]
        _lib = L
    return _lib


MODES = ("reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "grid-wrap", "wrap")


def _spline(mat, map_kind, order, mode, xcenter=0.0, ycenter=0.0, list_fact=(), list_coef=None, ycoord=None,
            xcoord=None, poly=1):
    """order 2..5: scipy's prefiltered spline interpolation (oracle/unwarp_oracle.c, spline section)."""
    mat = _f32c(mat)
    (height, width) = mat.shape
    m = MODES.index(mode)
    pad = lib().orc_spline_pad(m)
    work = np.empty((height + 2 * pad, width + 2 * pad), np.float64)
    f = _facts(list_fact)
    c = _facts(list_coef if list_coef is not None else [0.0] * 8)
    if map_kind == 2:
        yc_, xc_ = np.ascontiguousarray(ycoord), np.ascontiguousarray(xcoord)
        if yc_.dtype != xc_.dtype or yc_.dtype not in (np.float32, np.float64):
            raise TypeError("coordinates must both be float32 or both float64")
        out = np.empty(yc_.shape, np.float32)
        yp, xp, is64, n = yc_.ctypes.data, xc_.ctypes.data, int(yc_.dtype == np.float64), yc_.size
    else:
        out = np.empty((height, width), np.float32)
        yp = xp = None
        is64, n = 0, 0
    _check(lib().orc_remap_spline_f32(_fp(mat), _fp(out), height, width, _row_stride(mat), map_kind, float(xcenter),
                                      float(ycenter), _dp(f), f.size, _dp(c), yp, xp, is64, n, int(order), m, poly,
                                      _dp(work)))
    return out


def spline_coefficients(mat, order, mode):
    mat = _f32c(mat)
    (height, width) = mat.shape
    m = MODES.index(mode)
    pad = lib().orc_spline_pad(m)
    work = np.empty((height + 2 * pad, width + 2 * pad), np.float64)
    _check(lib().orc_spline_coefficients_f32(_fp(mat), height, width, _row_stride(mat), int(order), m, _dp(work)))
    return work


# element types of the typed path (codes shared with DCP_DTYPE_* of include/discorpy_hip.h)
DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3,
          np.dtype(np.uint16): 4, np.dtype(np.int16): 5, np.dtype(np.uint32): 6, np.dtype(np.int32): 7,
          np.dtype(np.int64): 8, np.dtype(np.uint64): 9, np.dtype(np.bool_): 10}


def map_coordinates(mat, ycoord, xcoord, order=1, mode="reflect"):
    """scipy.ndimage.map_coordinates(mat, (ycoord, xcoord), order, mode) for a 2-D `mat` of any of the
    DTYPES and coordinates clamped into the image; output has mat's dtype and ycoord's shape."""
    mat = np.asarray(mat)
    if mat.dtype not in DTYPES:
        raise TypeError("oracle does not handle dtype %s" % mat.dtype)
    if mat.ndim != 2 or mat.strides[1] != mat.itemsize or mat.strides[0] % mat.itemsize:
        mat = np.ascontiguousarray(mat)
    ycoord, xcoord = np.ascontiguousarray(ycoord), np.ascontiguousarray(xcoord)
    if ycoord.dtype != xcoord.dtype or ycoord.dtype not in (np.float32, np.float64):
        raise TypeError("coordinates must both be float32 or both float64")
    (height, width) = mat.shape
    m = MODES.index(mode)
    pad = lib().orc_spline_pad(m)
    work = np.empty((height + 2 * pad, width + 2 * pad) if int(order) >= 2 else (1,), np.float64)
    out = np.empty(ycoord.shape, mat.dtype)
    _check(lib().orc_map_coordinates_typed(mat.ctypes.data, out.ctypes.data, DTYPES[mat.dtype], height, width,
                                           mat.strides[0] // mat.itemsize, ycoord.ctypes.data, xcoord.ctypes.data,
                                           int(ycoord.dtype == np.float64), ycoord.size, int(order), m, _dp(work)))
    return out


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


def _f32c(a):
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise TypeError("oracle handles float32 data only")
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _facts(list_fact):
    return np.ascontiguousarray(np.asarray(list_fact, dtype=np.float64).reshape(-1))


def _check(rc):
    if rc != 0:
        raise ValueError("oracle rejected its arguments (rc=%d)" % rc)


def _row_stride(mat):
    if mat.ndim != 2 or mat.strides[1] != mat.itemsize or mat.strides[0] % mat.itemsize:
        raise ValueError("need unit column stride")
    return mat.strides[0] // mat.itemsize


def radial_coords(height, width, xcenter, ycenter, list_fact, poly=POLY_NUMPY, round_f32=True):
    """(yd, xd) float64 planes of postprocessing.py:138-145."""
    f = _facts(list_fact)
    yd = np.empty((height, width), np.float64)
    xd = np.empty((height, width), np.float64)
    _check(lib().orc_radial_coords(height, width, float(xcenter), float(ycenter), _dp(f), f.size,
                                   poly, int(round_f32), _dp(yd), _dp(xd)))
    return yd, xd


def perspective_coords(height, width, list_coef, round_f32=True):
    c = _facts(list_coef)
    yd = np.empty((height, width), np.float64)
    xd = np.empty((height, width), np.float64)
    _check(lib().orc_perspective_coords(height, width, _dp(c), int(round_f32), _dp(yd), _dp(xd)))
    return yd, xd


def unwarp_image_backward(mat, xcenter, ycenter, list_fact, order=1, mode="reflect", *,
                          poly=POLY_NUMPY, blend=BLEND_SCIPY, coord_round_f32=True):
    """postprocessing.py:111-148 for float32 `mat`, order 0/1 (mode is inert there)."""
    if np.asarray(mat).dtype != np.float32:
        (height, width) = np.shape(mat)
        yd, xd = radial_coords(height, width, xcenter, ycenter, list_fact, poly=poly, round_f32=coord_round_f32)
        return map_coordinates(mat, yd, xd, order, mode)
    if int(order) >= 2:
        return _spline(mat, 0, order, mode, xcenter, ycenter, list_fact, poly=poly)
    mat = _f32c(mat)
    (height, width) = mat.shape
    f = _facts(list_fact)
    out = np.empty((height, width), np.float32)
    _check(lib().orc_unwarp_image_f32(_fp(mat), _fp(out), height, width, _row_stride(mat),
                                      float(xcenter), float(ycenter), _dp(f), f.size, int(order),
                                      int(coord_round_f32), poly, blend))
    return out


def correct_perspective_image(mat, list_coef, order=1, mode="reflect", map_index=None, *,
                              blend=BLEND_SCIPY):
    """postprocessing.py:462-492."""
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    if np.asarray(mat).dtype != np.float32:
        (height, width) = np.shape(mat)
        if map_index is None:
            yd, xd = perspective_coords(height, width, list_coef)
        else:
            yd, xd = (np.ascontiguousarray(np.asarray(m).reshape(-1)) for m in map_index)
        return map_coordinates(mat, yd, xd, order, mode).reshape(height, width)
    mat = _f32c(mat)
    (height, width) = mat.shape
    if int(order) >= 2:
        if map_index is None:
            return _spline(mat, 1, order, mode, list_coef=list_coef)
        ycoord, xcoord = (np.ascontiguousarray(np.asarray(m).reshape(-1)) for m in map_index)
        return _spline(mat, 2, order, mode, ycoord=ycoord, xcoord=xcoord).reshape(height, width)
    out = np.empty((height, width), np.float32)
    if map_index is None:
        c = _facts(list_coef)
        _check(lib().orc_perspective_image_f32(_fp(mat), _fp(out), height, width, _row_stride(mat),
                                               _dp(c), int(order), blend))
        return out
    ycoord, xcoord = (np.ascontiguousarray(np.asarray(m).reshape(-1)) for m in map_index)
    return remap_coords(mat, ycoord, xcoord, order=order, blend=blend).reshape(height, width)


def remap_coords(mat, ycoord, xcoord, order=1, *, blend=BLEND_SCIPY, mode="reflect"):
    """map_coordinates(mat, (ycoord, xcoord), order, mode): coordinates outside the image follow scipy's `mode` at
    orders 0 and 1 ('nearest' clamps them) and are clamped at the spline orders."""
    if np.asarray(mat).dtype != np.float32:
        return map_coordinates(mat, ycoord, xcoord, order, mode)
    if int(order) >= 2:
        return _spline(mat, 2, order, mode, ycoord=ycoord, xcoord=xcoord)
    mat = _f32c(mat)
    ycoord = np.ascontiguousarray(ycoord)
    xcoord = np.ascontiguousarray(xcoord)
    if ycoord.dtype != xcoord.dtype or ycoord.dtype not in (np.float32, np.float64):
        raise TypeError("coordinates must both be float32 or both float64")
    out = np.empty(ycoord.shape, np.float32)
    _check(lib().orc_remap_coords_f32(_fp(mat), _fp(out), mat.shape[0], mat.shape[1], _row_stride(mat),
                                      ycoord.ctypes.data, xcoord.ctypes.data,
                                      int(ycoord.dtype == np.float64), ycoord.size, int(order), blend, MODES.index(mode)))
    return out


def unwarp_fused(mat, xcenter, ycenter, list_fact, list_coef, order=1, *, poly=POLY_NUMPY,
                 blend=BLEND_SCIPY):
    """One-pass perspective->radial remap (SURVEY.md section 8(d) cfg3 definition)."""
    if len(list_coef) != 8:
        raise ValueError("!!! Eight coefficients are required !!!")
    mat = _f32c(mat)
    (height, width) = mat.shape
    f, c = _facts(list_fact), _facts(list_coef)
    out = np.empty((height, width), np.float32)
    _check(lib().orc_unwarp_fused_f32(_fp(mat), _fp(out), height, width, _row_stride(mat),
                                      float(xcenter), float(ycenter), _dp(f), f.size, _dp(c),
                                      int(order), poly, blend))
    return out


def unwarp_stack_rows(mat3D, xcenter, ycenter, list_fact, row_start, nrows, *, coord_round_f32,
                      poly=POLY_NUMPY, blend=BLEND_SCIPY):
    if np.asarray(mat3D).dtype != np.float32:
        # any other element type: the coordinates of the requested rows once (x runs fastest, as :216-225 /
        # :303-309 build them), then one map_coordinates per projection in the input's dtype (:226-228, :310-312)
        mat3D = np.asarray(mat3D)
        (depth, height, width) = mat3D.shape
        f = _facts(list_fact)
        yd = np.empty((nrows, width), np.float64)
        xd = np.empty((nrows, width), np.float64)
        _check(lib().orc_radial_coords_rows(height, width, float(xcenter), float(ycenter), _dp(f), f.size, poly,
                                            int(coord_round_f32), float(row_start), nrows, _dp(yd), _dp(xd)))
        out = np.empty((depth, nrows, width), mat3D.dtype)
        b0, b1 = 0, height
        if coord_round_f32 and nrows > 0:
            # the chunk function crops the band [yd_min, yd_max) and scipy reflects inside it (:289-312)
            f = _facts(list_fact)
            lo, hi = C.c_int64(0), C.c_int64(0)
            lib().orc_chunk_band(height, width, float(xcenter), float(ycenter), _dp(f), f.size, float(row_start),
                                 float(row_start + nrows - 1), C.byref(lo), C.byref(hi))
            b0, b1 = int(lo.value), int(hi.value)
        # (yd_mat - yd_min is a float32 subtraction in the chunk function, :305-307; the slice function subtracts in float64)
        yrel = (yd.astype(np.float32) - np.float32(b0)) if coord_round_f32 else yd - b0
        for d in range(depth):
            out[d] = map_coordinates(mat3D[d, b0:b1], yrel, xd.astype(np.float32) if coord_round_f32 else xd, 1, "reflect")
        return out
    mat3D = np.ascontiguousarray(_f32c(mat3D))
    (depth, height, width) = mat3D.shape
    f = _facts(list_fact)
    out = np.empty((depth, nrows, width), np.float32)
    _check(lib().orc_unwarp_stack_rows_f32(_fp(mat3D), _fp(out), depth, height, width, float(xcenter),
                                           float(ycenter), _dp(f), f.size, float(row_start), nrows,
                                           int(coord_round_f32), poly, blend))
    return out


def unwarp_slice_backward(mat3D, xcenter, ycenter, list_fact, index, *, poly=POLY_NUMPY,
                          blend=BLEND_SCIPY):
    """postprocessing.py:188-229 (float64 coordinates, float32 output)."""
    if len(np.shape(mat3D)) < 3:
        raise ValueError("Input must be a 3D data")
    # :224 allocates the sinogram as float32 whatever the input dtype; each row is first produced in the
    # input's dtype by map_coordinates (:227) and converted by the assignment
    return unwarp_stack_rows(mat3D, xcenter, ycenter, list_fact, index, 1, coord_round_f32=False,
                             poly=poly, blend=blend)[:, 0, :].astype(np.float32, copy=False)


def unwarp_chunk_slices_backward(mat3D, xcenter, ycenter, list_fact, start_index, stop_index, *,
                                 poly=POLY_NUMPY, blend=BLEND_SCIPY):
    """postprocessing.py:255-313 (float32 coordinates, rows start..stop inclusive)."""
    if len(np.shape(mat3D)) < 3:
        raise ValueError("Input must be a 3D data")
    height = np.shape(mat3D)[1]
    index_list = np.arange(height, dtype=np.int16)
    if stop_index == -1:
        stop_index = height
    if (start_index not in index_list) or (stop_index not in index_list):
        raise ValueError("Selected index is out of the range")
    return unwarp_stack_rows(mat3D, xcenter, ycenter, list_fact, start_index,
                             stop_index - start_index + 1, coord_round_f32=True, poly=poly, blend=blend)


def correct_perspective_line(list_lines, list_coef):
    """The homography applied to lists of (y, x) points, numpy's operation order (reference postprocessing.py:414-441)."""
    c1, c2, c3, c4, c5, c6, c7, c8 = [float(v) for v in list_coef]
    out = []
    for line in list_lines:
        line = np.asarray(line, dtype=np.float64)
        x, y = line[:, 1], line[:, 0]
        den = c7 * x + c8 * y + 1.0
        out.append(np.column_stack([(c4 * x + c5 * y + c6) / den, (c1 * x + c2 * y + c3) / den]))
    return out
