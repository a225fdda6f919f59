"""Grid search over the centre of distortion in one call: dcp_unwarp_stack_rows_centres_f32 /
post.unwarp_slice_backward_centres (stack_centres_kernel).  The reference's pattern: examples/example_05.py:62-65 calls
unwarp_slice_backward (postprocessing.py:188-229) 121 times on one stack, one call per candidate centre.  Every centre's block
must equal the single call -- and the oracle -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import DEV, HOST, golden, noise, oblend

from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp


def _grid(xc, yc, half, step):
    return [(xc + dx, yc + dy) for dx in range(-half, half + step, step) for dy in range(-half, half + step, step)]


def test_centres_front_end_validation_needs_no_gpu():
    vol = np.zeros((2, 8, 9), np.float32)
    with pytest.raises(ValueError, match="Input must be a 3D data"):
        pp.unwarp_slice_backward_centres(vol[0], [1.0], [1.0], [1.0], 3)
    with pytest.raises(ValueError):
        pp.unwarp_slice_backward_centres(vol, [1.0, 2.0], [1.0], [1.0], 3)
    with pytest.raises(ValueError, match="Selected index is out of the range"):
        pp.unwarp_chunk_slices_backward_centres(vol, [1.0], [1.0], [1.0], 2, 8)
    L = F.lib()
    one = (C.c_double * 1)(1.0)
    assert L.dcp_unwarp_stack_rows_centres_f32(vol.ctypes.data, vol.ctypes.data, 2, 8, 9, 72, 9, one, one, -1, one, 1, 0.0, 1, 0, F.BLEND_F64LERP,
                                               F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_stack_rows_centres_f32(vol.ctypes.data, vol.ctypes.data, 2, 8, 9, 72, 9, None, one, 1, one, 1, 0.0, 1, 0, F.BLEND_F64LERP,
                                               F.MEM_HOST, -1, None) == F.ERR_INVALID_ARG
    assert L.dcp_unwarp_stack_rows_centres_f32(vol.ctypes.data, vol.ctypes.data, 2, 8, 9, 72, 9, one, one, 0, one, 1, 0.0, 1, 0, F.BLEND_F64LERP,
                                               F.MEM_HOST, -1, None) == F.OK


@pytest.mark.gpu
@pytest.mark.parametrize("blend", ["f64lerp", "scipy"])
def test_slice_grid_search_equals_single_calls_and_oracle(hip, orc, blend):
    g = golden("g6_stack3x800x1280")
    vol = noise(g["seed"], g["shape"])
    xc, yc, fact = float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"])
    cents = _grid(xc, yc, 40, 20)                        # 5 x 5 candidates
    xs, ys = [c[0] for c in cents], [c[1] for c in cents]
    ob = orc.BLEND_SCIPY if blend == "scipy" else orc.BLEND_F64LERP
    for index in (14, 400, 799, 14.5, -3):               # the reference does not validate `index` (golden G18)
        got = pp.unwarp_slice_backward_centres(vol, xs, ys, fact, index, blend=blend)
        assert got.shape == (len(cents), 3, 1280) and got.dtype == np.float32
        for k, (cx, cy) in enumerate(cents):
            assert np.array_equal(got[k], pp.unwarp_slice_backward(vol, cx, cy, fact, index, blend=blend)), (index, k)
            assert np.array_equal(got[k], orc.unwarp_slice_backward(vol, cx, cy, fact, index, poly=orc.POLY_KERNEL, blend=ob)), (index, k)
    # the golden centre itself: the reference's own output (G6)
    got = pp.unwarp_slice_backward_centres(vol, [xc + 1.0, xc], [yc, yc], fact, 14, blend="scipy")
    assert got.shape == (2, 3, 1280) and np.array_equal(got[1], g["slice_14"])
    got = pp.unwarp_chunk_slices_backward_centres(vol, [xc + 1.0, xc], [yc, yc], fact, 395, 402, blend="scipy")
    assert got.shape == (2, 3, 8, 1280) and np.array_equal(got[1], g["chunk_395_402"])


@pytest.mark.gpu
def test_device_stack_many_centres_and_chunks(hip, orc):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    D, H, W = 5, 300, 700
    vol = noise(11, (D, H, W))
    fact = [1.0, -3e-5, 4e-8]
    cents = [(350.0 + 0.37 * i, 150.0 - 0.21 * i) for i in range(230)]      # 230 > the 224 one launch carries
    xs, ys = [c[0] for c in cents], [c[1] for c in cents]
    t = torch.from_numpy(vol).cuda()
    got = pp.unwarp_slice_backward_centres(t, xs, ys, fact, 120)
    torch.cuda.synchronize()
    assert F.last_kernel().startswith("stack_centres_kernel") and tuple(got.shape) == (230, D, W)
    got = got.cpu().numpy()
    for k in (0, 1, 100, 223, 224, 229):
        assert np.array_equal(got[k], orc.unwarp_slice_backward(vol, xs[k], ys[k], fact, 120, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP)), k
    # chunk of rows (float32 coordinates), device and host
    got = pp.unwarp_chunk_slices_backward_centres(t, xs[:7], ys[:7], fact, 100, 131)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    host = pp.unwarp_chunk_slices_backward_centres(vol, xs[:7], ys[:7], fact, 100, 131, blend=DEV)
    assert got.shape == host.shape == (7, D, 32, W) and np.array_equal(got, host)
    for k in range(7):
        assert np.array_equal(got[k], orc.unwarp_chunk_slices_backward(vol, xs[k], ys[k], fact, 100, 131, poly=orc.POLY_KERNEL, blend=orc.BLEND_F64LERP))


@pytest.mark.gpu
def test_centres_folding_model_other_dtypes_and_strided_host_stack(hip, orc):
    g = golden("g15_folding_chunk")                     # a model that folds rows out of the reference's band (golden G15)
    fvol = noise(g["seed"], g["shape"])
    fold, fx, fy = list(g["list_fact"]), float(g["xcenter"]), float(g["ycenter"])
    s0, s1 = int(g["start"]), int(g["stop"])
    got = pp.unwarp_chunk_slices_backward_centres(fvol, [fx + 2.0, fx], [fy - 1.0, fy], fold, s0, s1, blend="scipy")
    assert np.array_equal(got[1], g["ref_out"])                            # the reference's own output, band reflection included
    assert np.array_equal(got[0], pp.unwarp_chunk_slices_backward(fvol, fx + 2.0, fy - 1.0, fold, s0, s1, blend="scipy"))
    # the slice function has no band to leave: batched even for that model
    got = pp.unwarp_slice_backward_centres(fvol, [fx + 2.0, fx], [fy - 1.0, fy], fold, s0 + 3)
    assert F.last_kernel().startswith("stack_centres_kernel")
    for k, (cx, cy) in enumerate([(fx + 2.0, fy - 1.0), (fx, fy)]):
        assert np.array_equal(got[k], orc.unwarp_slice_backward(fvol, cx, cy, fold, s0 + 3, poly=orc.POLY_KERNEL, blend=oblend(orc, HOST)))
    D, H, W = 4, 120, 200
    vol = noise(12, (D, H, W))
    cents = [(100.0, 60.0), (104.0, 57.5), (93.0, 64.0)]
    xs, ys = [c[0] for c in cents], [c[1] for c in cents]
    # uint16 stacks: centre by centre through the typed path, float32 result of the value rounded to uint16 (reference :224-227)
    u16 = (vol * 60000).astype(np.uint16)
    fact = [1.0, -3e-5, 4e-8]
    got = pp.unwarp_slice_backward_centres(u16, xs, ys, fact, 50)
    assert got.dtype == np.float32
    for k, (cx, cy) in enumerate(cents):
        assert np.array_equal(got[k], pp.unwarp_slice_backward(u16, cx, cy, fact, 50))
    # a host stack with padded rows and projections (strides through the C ABI)
    big = noise(13, (D, H + 3, W + 8))
    view = big[:, :H, :W]
    got = pp.unwarp_slice_backward_centres(view, xs, ys, fact, 77)
    for k, (cx, cy) in enumerate(cents):
        assert np.array_equal(got[k], orc.unwarp_slice_backward(np.ascontiguousarray(view), cx, cy, fact, 77, poly=orc.POLY_KERNEL,
                                                                blend=oblend(orc, HOST)))
