"""CPU suite, part 2: the C ABI loads and exports what include/discorpy_hip.h declares; the Python
front end validates like the reference (same exception types and messages) before touching a GPU.
No compute call is made here."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "discorpy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcp_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(F.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.realpath(F.LIB_PATH).startswith(os.path.realpath(ROOT))


def test_every_declared_symbol_is_exported_and_bound():
    lib = F.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
        assert n in F.SIGNATURES, "%s has no ctypes signature in _ffi.SIGNATURES" % n
    assert sorted(F.SIGNATURES) == names


def test_version_and_device_count_do_not_need_a_gpu():
    assert F.lib().dcp_version() >= 100
    assert F.device_count() >= 0


def test_options_round_trip():
    # the documented options by their names, the lab's switches by theirs with the "x_" prefix -- and neither the other way round
    for key, val in [("x_tile_rows", 8), ("x_pipe_depth", 4), ("x_xcd_remap", 1), ("x_xcd_remap", 2), ("x_coef_lds", 1), ("x_d_chunk", 32),
                     ("lds_gather", 1), ("x_stack_lds", 2), ("stack_chunk_kb", 512), ("host_duplex", 0), ("x_int_exact", 0), ("host_direct", 2),
                     ("x_box_table", 0), ("x_store_wait", 0), ("x_tall_tiles", 1), ("x_wg_per_cu", 2), ("x_stack_wg", 2), ("x_spline_tiled", 3),
                     ("x_fused_wg", 0), ("host_bands", 4), ("tile_cert", 0)]:
        old = F.get_option(key)
        F.set_option(key, val)
        assert F.get_option(key) == val
        F.set_option(key, old)
    with pytest.raises(ValueError, match="unknown option"):
        F.set_option("no_such_knob", 1)
    for key in ("fused_wg", "any_order", "pf2d_chunk", "x_host_duplex", "x_tile_cert", "x_", ""):
        with pytest.raises(ValueError, match="unknown option"):
            F.set_option(key, 1)
        with pytest.raises(ValueError, match="unknown option"):
            F.get_option(key)
    # the unprefixed spellings the round-4 header documented stay accepted as deprecated aliases of the x_ names
    for key, val in (("tile_rows", 8), ("wg_box", 0), ("stack_wg", 2), ("int_exact", 0), ("spline_tiled", 3), ("xcd_remap", 1)):
        old = F.get_option("x_" + key)
        F.set_option(key, val)
        assert F.get_option("x_" + key) == val and F.get_option(key) == val
        F.set_option("x_" + key, old)
    # the header documents the stable options and nothing else of them
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "discorpy_hip.h")).read()
    doc = hdr[hdr.index("/* Options (process-wide)"):hdr.index("int dcp_set_option")]
    import re as _re
    assert sorted(set(_re.findall(r'^ \*   "([a-z_]+)"', doc, _re.M))) == sorted(
        ["stack_chunk_kb", "host_duplex", "host_bands", "host_direct", "host_direct_applies", "tile_cert", "lds_gather"])
    with pytest.raises(ValueError):
        F.set_option("x_tile_rows", 1000)
    with pytest.raises(ValueError):
        F.set_option("x_pipe_depth", 3)


def test_abi_rejects_bad_arguments_before_any_gpu_work():
    L = F.lib()
    fa, n = F.fact_array([1.0, 1e-3])
    buf = np.zeros((4, 4), np.float32)
    p = buf.ctypes.data
    bad = [
        L.dcp_unwarp_image_f32(None, p, 4, 4, 4, 1, 0.0, 0.0, fa, n, 1, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_unwarp_image_f32(p, p, 0, 4, 4, 1, 0.0, 0.0, fa, n, 1, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_unwarp_image_f32(p, p, 4, 4, 2, 1, 0.0, 0.0, fa, n, 1, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_unwarp_image_f32(p, p, 4, 4, 4, 1, 0.0, 0.0, fa, 33, 1, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_unwarp_image_f32(p, p, 4, 4, 4, 1, 0.0, 0.0, fa, n, 1, 1, 7, F.MEM_HOST, -1, None),
        L.dcp_perspective_image_f32(p, p, 4, 4, 4, 1, None, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_remap_coords_f32(p, p, 4, 4, 4, 1, p, p, 5, 16, 1, 0, F.MEM_HOST, -1, None),
        L.dcp_remap_coords_mode_f32(p, p, 4, 4, 4, 1, p, p, 0, 16, 1, 8, 0, F.MEM_HOST, -1, None),     # boundary mode 8
        L.dcp_remap_coords_mode_f32(p, p, 4, 4, 4, 1, p, p, 0, -1, 1, 0, 0, F.MEM_HOST, -1, None),     # npts < 0
        L.dcp_unwarp_stack_rows_f32(p, p, 1, 4, 4, 16, 2, 0.0, 0.0, fa, n, 0.0, 1, 1, 0, F.MEM_HOST, -1, None),
    ]
    assert all(rc == F.ERR_INVALID_ARG for rc in bad), bad
    assert "blend_mode" in F.last_error() or len(F.last_error()) > 0
    assert L.dcp_unwarp_image_f32(p, p, 4, 4, 4, 1, 0.0, 0.0, fa, n, 3, 1, 0, F.MEM_HOST, -1, None) == F.ERR_UNSUPPORTED
    with pytest.raises(NotImplementedError, match="order 3"):
        F.check(F.ERR_UNSUPPORTED)
    assert L.dcp_unwarp_image_spline_f32(p, p, 4, 4, 4, 1, 0.0, 0.0, fa, n, 1, 0, F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_image_spline_f32(p, p, 4, 4, 4, 1, 0.0, 0.0, fa, n, 3, 9, F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG


# ---- reference error behaviour, raised on the host before the device is needed

def test_perspective_needs_eight_coefficients():
    with pytest.raises(ValueError, match="!!! Eight coefficients are required !!!"):
        pp.correct_perspective_image(np.zeros((4, 4), np.float32), [1.0] * 7)
    with pytest.raises(ValueError, match="!!! Eight coefficients are required !!!"):
        pp.unwarp_perspective_fused(np.zeros((4, 4), np.float32), 1, 1, [1.0], [1.0] * 9)


def test_stack_functions_need_3d_input():
    flat = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError, match="Input must be a 3D data"):
        pp.unwarp_slice_backward(flat, 4, 4, [1.0], 2)
    with pytest.raises(ValueError, match="Input must be a 3D data"):
        pp.unwarp_chunk_slices_backward(flat, 4, 4, [1.0], 1, 2)


def test_chunk_index_validation_matches_the_reference():
    vol = np.zeros((2, 8, 8), np.float32)
    for start, stop in [(-1, 3), (0, 8), (3, 99), (0, -1), (2.5, 4)]:
        with pytest.raises(ValueError, match="Selected index is out of the range"):
            pp.unwarp_chunk_slices_backward(vol, 4, 4, [1.0], start, stop)
    # start beyond stop: the reference maps zero rows and returns an empty (depth, 0, width) array of the input's type
    for dt in (np.float32, np.uint16):
        empty = pp.unwarp_chunk_slices_backward(vol.astype(dt), 4, 4, [1.0], 5, 0)
        assert empty.shape == (2, 0, 8) and empty.dtype == dt


def test_image_shape_errors_surface_like_the_reference():
    with pytest.raises(ValueError, match="too many values to unpack"):
        pp.unwarp_image_backward(np.zeros((2, 3, 4), np.float32), 1, 1, [1.0])
    with pytest.raises(AttributeError):
        pp.unwarp_image_backward([[1.0, 2.0]], 1, 1, [1.0])


def test_unsupported_inputs_fail_loudly_instead_of_falling_back():
    img = np.zeros((4, 4), np.float32)
    with pytest.raises(RuntimeError, match="spline order not supported"):
        pp.unwarp_image_backward(img, 1, 1, [1.0], order=6)
    # every element type scipy takes has a code now (64-bit integers, bool; complex goes through as two real arrays); what
    # scipy refuses is refused with its words
    for dt in (np.int64, np.uint64, np.bool_):
        assert pp._dtype_code(np.dtype(dt)) in (8, 9, 10)
    assert pp._complex_parts(np.zeros((2, 2), np.complex64)) is not None and pp._complex_parts(img) is None
    with pytest.raises(RuntimeError, match="data type not supported"):      # scipy's own refusal of float16, word for word
        pp.unwarp_image_backward(img.astype(np.float16), 1, 1, [1.0])
    with pytest.raises(TypeError):                                           # scipy wants an integer order
        pp.unwarp_image_backward(img, 1, 1, [1.0], order=1.0)
    with pytest.raises(TypeError):                                           # index - ycenter in the reference
        pp.unwarp_slice_backward(np.zeros((2, 4, 4), np.float32), 1, 1, [1.0], "a")
    with pytest.raises(ValueError):                                          # map_index arrays of different sizes
        pp.correct_perspective_image(img, [1, 0, 0, 0, 1, 0, 0, 0], map_index=(np.zeros(16, np.float32), np.zeros(12, np.float32)))
    with pytest.raises(RuntimeError, match="boundary mode not supported"):
        pp.unwarp_image_backward(img, 1, 1, [1.0], mode="bogus")
    with pytest.raises(ValueError, match="unknown blend"):
        pp.unwarp_image_backward(img, 1, 1, [1.0], blend="bf16")


def test_no_silent_cpu_path(monkeypatch):
    """With no device the product path must raise -- it must never route through the oracle."""
    monkeypatch.setattr(F, "device_count", lambda: 0)
    with pytest.raises(F.HipError, match="no CPU fallback"):
        pp.unwarp_image_backward(np.zeros((4, 4), np.float32), 1, 1, [1.0])
    import discorpy_amd
    pkg = os.path.dirname(discorpy_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                text = open(path).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path
                assert "libunwarp_oracle" not in text, path
                # nor may the product lean on the reference itself (utility.py once imported discorpy.proc)
                assert not re.search(r"^\s*(import|from)\s+discorpy(\.|\s|$)", text, flags=re.M), path
                code = re.sub(r"#[^\n]*", "", re.sub(r'"""[\s\S]*?"""', "", text))     # citations in comments are fine
                assert "/root/reference" not in code, path
            elif f.endswith((".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(path).read()
                assert not re.search(r"#\s*include[^\n]*oracle", text), path
                assert "libunwarp_oracle" not in text and "-lunwarp_oracle" not in text, path


def test_coefficient_file_of_the_reference_parses():
    """data/coef_dot_05.txt format ('key : value' per line, loadersaver.py:768-776) -> configs constants."""
    from discorpy_amd import configs
    text = ("xcenter : 588.692801577\nycenter : 462.092631791\nfactor0 : 1.00227490554\n"
            "factor1 : -2.99523692178e-05\nfactor2 : 8.99519088e-08\nfactor3 : -1.57066461911e-10\n"
            "factor4 : 8.08880211618e-14\n")
    vals = [float(line.split()[-1]) for line in text.splitlines()]
    assert vals[0] == configs.XCENTER_DOT_05 and vals[1] == configs.YCENTER_DOT_05
    assert tuple(vals[2:]) == configs.COEF_DOT_05


def test_host_output_pool_recycles_only_unreachable_blocks():
    """discorpy_amd/_pool.py: a block returns to the pool when the caller has dropped the array and every view of it,
    never earlier; small outputs and a disabled pool are plain np.empty."""
    from discorpy_amd import _pool
    _pool.clear()
    a = _pool.empty((512, 1024), np.float32)
    assert a.shape == (512, 1024) and a.dtype == np.float32 and a.flags.writeable and a.flags.c_contiguous
    a[:] = 3.0
    addr = a.ctypes.data
    view = a[100:200, ::2]
    del a
    assert _pool.stats()["idle_bytes"] == 0                  # the view keeps the block alive
    b = _pool.empty((512, 1024), np.float32)
    assert b.ctypes.data != addr
    assert float(view[0, 0]) == 3.0
    del view
    assert _pool.stats()["idle_bytes"] == 512 * 1024 * 4
    c = _pool.empty((1024, 512), np.float32)                 # same byte count, other shape: reused
    assert c.ctypes.data == addr and _pool.stats()["idle_bytes"] == 0
    d = _pool.empty((512, 1024), np.uint16)                  # other size: a new block
    assert d.ctypes.data not in (addr, b.ctypes.data)
    small = _pool.empty((10, 10), np.float32)
    assert small.flags.owndata
    off = _pool.HostPool(0).empty((512, 1024), np.float32)
    assert off.flags.owndata
    small_pool = _pool.HostPool(10 << 20)                      # beyond the cap the longest-idle blocks go first
    blocks = [small_pool.empty((1 << 20,), np.float32) for _ in range(4)]
    addrs = [blk.ctypes.data for blk in blocks]
    del blocks
    assert small_pool.idle_bytes == 8 << 20 and len(small_pool.age) == 2
    again = small_pool.empty((1 << 20,), np.float32)
    assert again.ctypes.data in addrs
    del b, c, d
    _pool.clear()
    assert _pool.stats()["idle_bytes"] == 0


def test_out_argument_is_validated_on_the_host(monkeypatch):
    img = np.zeros((8, 8), np.float32)
    for bad in (np.zeros((8, 9), np.float32), np.zeros((8, 8), np.float64), np.zeros((8, 16), np.float32)[:, ::2], [[0.0]]):
        with pytest.raises(ValueError, match="out must be"):
            pp.unwarp_image_backward(img, 4, 4, [1.0], out=bad)
    with pytest.raises(ValueError, match="out must"):
        pp.unwarp_slice_backward(np.zeros((2, 8, 8), np.float32), 4, 4, [1.0], 3, out=np.zeros((2, 9), np.float32))


def test_stack_row_band_and_band_validation_need_no_gpu():
    """dcp_stack_row_band is host arithmetic; dcp_unwarp_stack_band rejects a band that does not cover the request
    before any device work."""
    from discorpy_amd import configs
    H, W = 800, 1280
    a = (configs.XCENTER_DOT_05, configs.YCENTER_DOT_05, list(configs.COEF_DOT_05))
    b0, bn = F.stack_row_band(H, W, *a, 400, 1)
    assert 380 <= b0 <= 400 and 2 <= bn <= 12
    b0c, bnc = F.stack_row_band(H, W, *a, 14, 7)                      # examples/example_04.py:92-96
    assert 0 <= b0c <= 14 and b0c + bnc >= 21 and bnc < 40
    assert F.stack_row_band(H, W, a[0], a[1], [], 10, 5) == (max(int(a[1]) - 1, 0), 4)   # B = 0: every row maps to yc
    full = F.stack_row_band(H, W, a[0], a[1], [float("nan")], 10, 5)
    assert full == (0, H)
    L = F.lib()
    fa, n = F.fact_array(a[2])
    buf = np.zeros((2, bn, W), np.float32)
    out = np.zeros((2, 1, W), np.float32)
    rc = L.dcp_unwarp_stack_band(buf.ctypes.data, out.ctypes.data, 0, 0, 2, H, W, b0 + 1, bn - 1, bn * W, W, a[0], a[1], fa, n,
                                 400.0, 1, 0, 1, F.MEM_HOST, -1, None)
    assert rc == F.ERR_INVALID_ARG and "band holds" in F.last_error()
    rc = L.dcp_unwarp_stack_band(buf.ctypes.data, out.ctypes.data, 0, 0, 2, H, W, 790, 20, bn * W, W, a[0], a[1], fa, n,
                                 400.0, 1, 0, 1, F.MEM_HOST, -1, None)
    assert rc == F.ERR_INVALID_ARG and "outside the projection" in F.last_error()
    with pytest.raises(ValueError):
        F.stack_row_band(H, W, *a, float("inf"), 1)


def test_non_native_byte_order_is_normalised():
    """HDF5 files often hold big-endian integers; scipy returns native-endian output for them and so does this path."""
    a = np.arange(12, dtype=">u2").reshape(3, 4)
    im = pp._Image(a, 2)
    assert im.code == F.DTYPE_BY_NAME["uint16"] and im.dtype == np.dtype("uint16") and im.dtype.isnative
    assert np.array_equal(im.keep, a) and im.keep.dtype.isnative


def test_out_arrays_that_cannot_be_viewed_are_refused_before_any_work():
    """out.reshape(...) of a non-contiguous `out` made a copy and the result silently landed there."""
    vol = np.zeros((4, 8, 6), np.float32)
    bad = np.zeros((4, 12), np.float32)[:, ::2]                    # (depth, width) but strided
    with pytest.raises(ValueError, match="C-contiguous"):
        pp.unwarp_slice_backward(vol, 3.0, 4.0, [1.0], 2, out=bad)
    bad4 = np.zeros((2, 4, 12), np.float32)[:, :, ::2]
    with pytest.raises(ValueError, match="C-contiguous"):
        pp.unwarp_slice_backward_centres(vol, [3.0, 3.5], [4.0, 4.5], [1.0], 2, out=bad4)
    # an empty chunk per centre, as the single-centre call (and the reference's np.arange of nothing) gives
    e = pp.unwarp_chunk_slices_backward_centres(vol, [3.0, 3.5], [4.0, 4.5], [1.0], 5, 4)
    assert e.shape == (2, 4, 0, 6) and e.dtype == np.float32
