"""ctypes binding of libdiscorpy_hip.so (C ABI: include/discorpy_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C discorpy_amd/csrc`` into
``discorpy_amd/lib/``.  There is no CPU fallback: if the library is missing or no HIP device is
visible, the entry points of :mod:`discorpy_amd.post.postprocessing` raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCP_LIB_PATH") or os.path.join(_HERE, "lib", "libdiscorpy_hip.so")

OK, ERR_INVALID_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = 0, -1, -2, -3, -4
MEM_HOST, MEM_DEVICE = 0, 1
MEM_DEVICE_UNORDERED = 0x101      # DCP_MEM_DEVICE_UNORDERED: device pointers, the launch may overlap the previous one of its stream
BLEND_SCIPY, BLEND_F64LERP, BLEND_F32LERP = 0, 1, 2
COORD_F32, COORD_F64 = 0, 1
COPY_H2D, COPY_D2H, COPY_D2D = 0, 1, 2
MAP_RADIAL, MAP_PERSPECTIVE, MAP_FUSED = 0, 1, 2
# DCP_DTYPE_* by NumPy / torch dtype name
DTYPE_BY_NAME = {"float32": 0, "float64": 1, "uint8": 2, "int8": 3, "uint16": 4, "int16": 5, "uint32": 6, "int32": 7,
                 "int64": 8, "uint64": 9, "bool": 10}
DTYPE_F32 = 0
MAX_FACT = 32

BLEND_BY_NAME = {"scipy": BLEND_SCIPY, "exact": BLEND_SCIPY, "f64": BLEND_F64LERP,
                 "f64lerp": BLEND_F64LERP, "f32": BLEND_F32LERP, "f32lerp": BLEND_F32LERP}


class HipLibraryMissing(ImportError):
    pass


class HipError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); mirrors include/discorpy_hip.h declaration by declaration
_i64, _dbl, _int, _vp, _sz = C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_size_t
_dp = C.POINTER(C.c_double)
SIGNATURES = {
    "dcp_version": (_int, []),
    "dcp_device_count": (_int, []),
    "dcp_last_error": (C.c_char_p, []),
    "dcp_release_scratch": (_int, []),
    "dcp_set_option": (_int, [C.c_char_p, _int]),
    "dcp_get_option": (_int, [C.c_char_p, C.POINTER(_int)]),
    "dcp_unwarp_image_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _int, _int,
                                    _int, _int, _int, _vp]),
    "dcp_unwarp_images_f32": (_int, [C.POINTER(_vp), C.POINTER(_vp), _int, _i64, _i64, _i64, _i64, _dp, _dp, _dp, _int, _int, _int,
                                     _int, _int, _int, _vp]),
    "dcp_perspective_image_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dp, _int, _int, _int, _int, _vp]),
    "dcp_unwarp_fused_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dp, _int, _int,
                                    _int, _int, _vp]),
    "dcp_remap_coords_mode_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _i64, _int, _int, _int, _int, _int, _vp]),
    "dcp_remap_coords_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _i64, _int, _int, _int,
                                    _int, _vp]),
    "dcp_unwarp_stack_rows_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dbl,
                                         _i64, _int, _int, _int, _int, _vp]),
    "dcp_unwarp_stack_rows_centres_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _dp, _dp, _int, _dp, _int, _dbl, _i64, _int, _int,
                                                 _int, _int, _vp]),
    "dcp_unwarp_stack_rows_multi_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dbl,
                                               _i64, _int, _int, C.POINTER(_int), _int]),
    "dcp_unwarp_stack_rows_peer_f32": (_int, [C.POINTER(_vp), C.POINTER(_vp), _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dbl,
                                              _i64, _int, _int, C.POINTER(_int), _int, _int]),
    "dcp_rccl_available": (_int, []),
    "dcp_rccl_unique_id": (_int, [_vp, _sz]),
    "dcp_rccl_comm_create": (_int, [C.POINTER(_vp), _int, _int, _vp, _int]),
    "dcp_rccl_comm_destroy": (_int, [_vp]),
    "dcp_rccl_comm_fixed_shards": (_int, [_vp, _int]),
    "dcp_rccl_comm_info": (_int, [_vp, C.POINTER(_i64), _int, C.POINTER(_i64), _int, C.c_char_p, _sz]),
    "dcp_unwarp_stack_rows_rccl_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dbl, _i64, _int, _int, _vp,
                                              _int, _vp]),
    "dcp_unwarp_image_spline_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _int, _int, _int,
                                           _int, _vp]),
    "dcp_perspective_image_spline_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dp, _int, _int, _int, _int, _vp]),
    "dcp_unwarp_fused_spline_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dp, _int, _int, _int,
                                           _int, _vp]),
    "dcp_remap_coords_spline_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _i64, _int, _int, _int,
                                           _int, _vp]),
    "dcp_unwarp_image_typed": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _int, _int, _int,
                                      _int, _vp]),
    "dcp_perspective_image_typed": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _dp, _int, _int, _int, _int, _vp]),
    "dcp_unwarp_fused_typed": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int, _dp, _int, _int,
                                      _int, _int, _vp]),
    "dcp_remap_coords_typed": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _vp, _vp, _int, _i64, _int, _int, _int,
                                      _int, _vp]),
    "dcp_unwarp_stack_rows_typed": (_int, [_vp, _vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp, _int,
                                           _dbl, _i64, _int, _int, _int, _vp]),
    "dcp_unwarp_image_channels": (_int, [_vp, _vp, _int, _i64, _i64, _int, _i64, _i64, _dbl, _dbl, _dp, _int, _int, _int,
                                         _int, _vp]),
    "dcp_unwarp_color_image": (_int, [_vp, _vp, _int, _i64, _i64, _int, _i64, _i64, _dbl, _dbl, _dp, _int, _int, _int, _int,
                                      _int, _vp]),
    "dcp_stack_row_band": (_int, [_i64, _i64, _dbl, _dbl, _dp, _int, _dbl, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "dcp_unwarp_stack_band": (_int, [_vp, _vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dp,
                                     _int, _dbl, _i64, _int, _int, _int, _int, _vp]),
    "dcp_map_points_f64": (_int, [_vp, _vp, _i64, _dbl, _dbl, _dp, _int, _int, _int, _vp]),
    "dcp_map_points_perspective_f64": (_int, [_vp, _vp, _i64, _dp, _int, _int, _vp]),
    "dcp_coordinate_map_f32": (_int, [_vp, _vp, _i64, _i64, _int, _dbl, _dbl, _dp, _int, _dp, _int, _int, _vp]),
    "dcp_debug_counters": (_int, [C.POINTER(C.c_uint64), _int, _int]),
    "dcp_debug_bounds": (_int, [C.POINTER(C.c_uint64), _int, _int]),
    "dcp_debug_last_kernel": (C.c_char_p, []),
    "dcp_debug_tile_certificate": (_int, [_int, _i64, _i64, _dbl, _dbl, _dp, _int, _dp]),
    "dcp_malloc": (_int, [C.POINTER(_vp), _sz, _int]),
    "dcp_free": (_int, [_vp, _int]),
    "dcp_memcpy": (_int, [_vp, _vp, _sz, _int, _int, _vp]),
    "dcp_host_register": (_int, [_vp, _sz, _int]),
    "dcp_host_unregister": (_int, [_vp]),
    "dcp_stream_create": (_int, [C.POINTER(_vp), _int]),
    "dcp_stream_destroy": (_int, [_vp]),
    "dcp_stream_synchronize": (_int, [_int, _vp]),
    "dcp_event_create": (_int, [C.POINTER(_vp), _int]),
    "dcp_event_record": (_int, [_vp, _vp]),
    "dcp_stream_wait_event": (_int, [_vp, _vp]),
    "dcp_event_synchronize": (_int, [_vp]),
    "dcp_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "dcp_event_destroy": (_int, [_vp]),
}


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not
    work (whichever initialises second sees no device), so when torch is installed its copy is
    loaded first -- without importing torch -- and libdiscorpy_hip.so, which only asks for the
    SONAME, binds to it.  Either import order then works.  DISCORPY_AMD_SYSTEM_HIP=1 opts out."""
    if os.environ.get("DISCORPY_AMD_SYSTEM_HIP") == "1":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        return cand
    except OSError:
        return None


def lib():
    """Load the shared library (once) and declare every entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "libdiscorpy_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C discorpy_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    msg = lib().dcp_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Translate a DCP_ERR_* code into the exception the reference's callers expect."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise HipError(msg)


def device_count():
    return int(lib().dcp_device_count())


def require_device():
    n = device_count()
    if n < 1:
        raise HipError("no HIP device visible: the discorpy_amd unwarp path runs on the GPU only "
                       "(no CPU fallback)")
    return n


def release_scratch():
    """Free the device scratch the library keeps between calls (this thread's staging, the spline planes) and the idle
    blocks of the host output pool."""
    from . import _pool
    _pool.clear()
    check(lib().dcp_release_scratch())


def set_option(key, value):
    check(lib().dcp_set_option(key.encode(), int(value)))


def get_option(key):
    v = C.c_int(0)
    check(lib().dcp_get_option(key.encode(), C.byref(v)))
    return v.value


def debug_counters(reset=True):
    """(tiles whose box did not fit, tiles whose containment vote failed) since the last reset."""
    out = (C.c_uint64 * 2)()
    check(lib().dcp_debug_counters(out, 2, int(reset)))
    return int(out[0]), int(out[1])


def debug_bounds(reset=True):
    """(taps outside their LDS slab, first offset, slab bytes, site, is_checking_build) -- see dcp_debug_bounds."""
    out = (C.c_uint64 * 5)()
    check(lib().dcp_debug_bounds(out, 5, int(reset)))
    return tuple(int(v) for v in out)


def last_kernel():
    """Name of the float32 image / stack kernel this thread launched last, e.g. 'remap_wg_kernel<Radial,NF=5,f64lerp>'."""
    v = lib().dcp_debug_last_kernel()
    return v.decode() if v else ""


def tile_certificate(height, width, xcenter, ycenter, list_fact=None, list_coef=None):
    """Level of the host's tile-deviation certificate (0, 1 or 2) for a radial (list_fact) or perspective (list_coef) map."""
    if list_coef is not None:
        ca, _ = fact_array(list_coef)
        rc = lib().dcp_debug_tile_certificate(MAP_PERSPECTIVE, int(height), int(width), 0.0, 0.0, None, 0, ca)
    else:
        fa, nf = fact_array(list_fact)
        rc = lib().dcp_debug_tile_certificate(MAP_RADIAL, int(height), int(width), float(xcenter), float(ycenter), fa, nf, None)
    if rc < 0:
        check(rc)
    return rc


def stack_row_band(height, width, xcenter, ycenter, list_fact, row_start, nrows):
    """(band_start, band_rows): source rows of a projection that output rows row_start .. row_start+nrows-1 can reach."""
    fa, nf = fact_array(list_fact)
    b0, bn = C.c_int64(0), C.c_int64(0)
    check(lib().dcp_stack_row_band(int(height), int(width), float(xcenter), float(ycenter), fa, nf, float(row_start),
                                   int(nrows), C.byref(b0), C.byref(bn)))
    return int(b0.value), int(bn.value)


def fact_array(list_fact):
    """(ctypes double array, n) for a coefficient sequence."""
    vals = [float(v) for v in list_fact]
    arr = (C.c_double * max(len(vals), 1))(*vals)
    return arr, len(vals)


class DeviceBuffer:
    """Device allocation owned by Python (dcp_malloc / dcp_free)."""

    def __init__(self, nbytes, device=-1):
        self.nbytes = int(nbytes)
        self.device = device
        p = C.c_void_p()
        check(lib().dcp_malloc(C.byref(p), self.nbytes, device))
        self.ptr = p.value

    def upload(self, array):
        import numpy as np
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        check(lib().dcp_memcpy(self.ptr, a.ctypes.data, a.nbytes, COPY_H2D, self.device, None))
        return self

    def download(self, shape, dtype):
        import numpy as np
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().dcp_memcpy(out.ctypes.data, self.ptr, out.nbytes, COPY_D2H, self.device, None))
        return out

    def free(self):
        if self.ptr:
            lib().dcp_free(self.ptr, self.device)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self, device=-1):
        p = C.c_void_p()
        check(lib().dcp_event_create(C.byref(p), device))
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().dcp_event_record(self.ptr, stream))

    def synchronize(self):
        check(lib().dcp_event_synchronize(self.ptr))

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        check(lib().dcp_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                lib().dcp_event_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class Stream:
    """A non-blocking HIP stream owned by Python (dcp_stream_create / dcp_stream_destroy)."""

    def __init__(self, device=-1):
        self.device = device
        p = C.c_void_p()
        check(lib().dcp_stream_create(C.byref(p), device))
        self.ptr = p.value

    def synchronize(self):
        check(lib().dcp_stream_synchronize(self.device, self.ptr))

    def wait_event(self, event):
        """Work enqueued on this stream from now on starts after `event` (an Event recorded on any stream) has completed."""
        check(lib().dcp_stream_wait_event(self.ptr, event.ptr))

    def __del__(self):
        try:
            if self.ptr:
                lib().dcp_stream_destroy(self.ptr)
                self.ptr = None
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


class DeviceArray:
    """Result of a call whose input was a device array that is not a torch tensor (CuPy, Numba, ... -- anything
    exposing ``__cuda_array_interface__``): owns a device allocation and exposes it through the same interface, so
    ``cupy.asarray(result)`` / ``torch.as_tensor(result, device="cuda")`` wrap it without a copy."""

    def __init__(self, shape, dtype, device=-1):
        import numpy as np
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape, dtype=np.int64))
        self.nbytes = self.size * self.dtype.itemsize
        self._buf = DeviceBuffer(max(self.nbytes, 4), device)
        self._offset = 0

    @property
    def ptr(self):
        return self._buf.ptr + self._offset

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3,
                "strides": None}

    def copy_to_host(self):
        import numpy as np
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().dcp_memcpy(out.ctypes.data, self.ptr, self.nbytes, COPY_D2H, self._buf.device, None))
        return out

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        """Block `i` along the first axis (an integer index only: a view of the same allocation)."""
        import operator
        i = operator.index(i)
        return self.frame(i + self.shape[0] if i < 0 else i)

    def copy_from_host(self, array):
        import numpy as np
        a = np.ascontiguousarray(array, dtype=self.dtype)
        if a.shape != self.shape:
            raise ValueError("expected shape %s" % (self.shape,))
        if a.nbytes:
            check(lib().dcp_memcpy(self.ptr, a.ctypes.data, a.nbytes, COPY_H2D, self._buf.device, None))
        return self

    def frame(self, i):
        """View of block `i` along the first axis (same allocation)."""
        import numpy as np
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        view = DeviceArray.__new__(DeviceArray)
        view.shape, view.dtype, view._buf = self.shape[1:], self.dtype, self._buf
        view.size = int(np.prod(view.shape, dtype=np.int64))
        view.nbytes = view.size * self.dtype.itemsize
        view._offset = self._offset + i * view.nbytes
        return view

    def reshape(self, shape):
        """A view with another shape (same allocation)."""
        import numpy as np
        view = DeviceArray.__new__(DeviceArray)
        view.shape = tuple(int(s) for s in np.empty(self.shape, np.bool_).reshape(shape).shape)
        view.dtype, view.size, view.nbytes, view._buf = self.dtype, self.size, self.nbytes, self._buf
        view._offset = self._offset
        return view
