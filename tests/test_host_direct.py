"""Host frames whose destination is REGISTERED host memory (the recycled NumPy outputs of the front end are): the kernels write
the result straight into it over PCIe, band by band, while the source is still uploading (csrc/api_image.cpp: run_host_direct).
The result must be what the staged path and the oracle give.  Reference: a NumPy array in, a NumPy array out
(postprocessing.py:111-148, 462-492)."""
import ctypes as C

import numpy as np
import pytest

from conftest import HOST, noise

from discorpy_amd import _ffi as F
from discorpy_amd import _pool
from discorpy_amd.post import postprocessing as pp


def test_register_entry_points_validate_without_a_gpu():
    L = F.lib()
    assert L.dcp_host_register(None, 10, -1) == F.ERR_INVALID_ARG
    assert L.dcp_host_unregister(None) == F.OK
    with pytest.raises(ValueError):
        F.set_option("host_direct", 3)
    a = _pool.empty((2048, 2048), np.float32)            # 16 MiB: would be registered on a GPU box; a plain block here or there, usable
    a[...] = 1.0
    assert float(a.sum()) == 2048.0 * 2048.0


@pytest.mark.gpu
def test_direct_write_into_registered_host_memory_equals_the_staged_path(hip, orc):
    H, W = 2200, 2100                                     # 18.5 MB: above the 16 MiB threshold of the direct / banded paths
    img = noise(501, (H, W))
    xc, yc, fact = 1010.4, 1120.7, [1.0, -8e-6, 6e-9, -2e-12]
    coef = [0.98, -0.012, 20.5, 0.009, 1.01, -14.0, 4e-6, -3e-6]
    hb = {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}[HOST]        # NumPy frames: the host default blend
    want = {"radial": orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL, blend=hb),
            "nearest": orc.unwarp_image_backward(img, xc, yc, fact, order=0, poly=orc.POLY_KERNEL),
            "persp": orc.correct_perspective_image(img, coef, blend=hb),
            "fused": orc.unwarp_fused(img, xc, yc, fact, coef, poly=orc.POLY_KERNEL, blend=hb)}
    L = hip.lib()
    try:
        for mode in (2, 0, 1):                            # always direct when registered / never / decided by the runtime probe
            hip.set_option("host_direct", mode)
            got = {"radial": pp.unwarp_image_backward(img, xc, yc, fact), "nearest": pp.unwarp_image_backward(img, xc, yc, fact, order=0),
                   "persp": pp.correct_perspective_image(img, coef), "fused": pp.unwarp_perspective_fused(img, xc, yc, fact, coef)}
            for k in want:
                assert np.array_equal(got[k], want[k]), (mode, k)
        # a caller's own registered buffer through the C ABI (out=): written in place
        hip.set_option("host_direct", 2)
        out = np.zeros((H, W), np.float32)
        hip.check(L.dcp_host_register(out.ctypes.data, out.nbytes, -1))
        try:
            res = pp.unwarp_image_backward(img, xc, yc, fact, out=out)
            assert res is out and np.array_equal(out, want["radial"])
            strided = np.zeros((H, W + 7), np.float32)
            strided[:, :W] = img
            res = pp.unwarp_image_backward(strided[:, :W], xc, yc, fact, out=out)       # padded source rows
            assert np.array_equal(out, want["radial"])
        finally:
            hip.check(L.dcp_host_unregister(out.ctypes.data))
    finally:
        hip.set_option("host_direct", 1)
    # the pool's blocks are registered and survive recycling
    del got, res
    a = pp.unwarp_image_backward(img, xc, yc, fact)
    addr = a.ctypes.data
    del a
    b = pp.unwarp_image_backward(img, xc, yc, fact)
    assert b.ctypes.data == addr and np.array_equal(b, want["radial"])
    _pool.clear()


@pytest.mark.gpu
def test_a_destination_registered_only_in_part_takes_the_staged_path(hip, orc):
    """the direct path asked about the FIRST byte of the destination only; a frame that starts inside a registered range
    shorter than itself would have been stored to through an alias that ends early."""
    H, W = 2200, 2100
    img = noise(502, (H, W))
    xc, yc, fact = 1010.4, 1120.7, [1.0, -8e-6, 6e-9, -2e-12]
    want = orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL,
                                     blend={"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}[HOST])
    L = hip.lib()
    out = np.zeros((H, W), np.float32)
    half = (out.nbytes // 2) & ~4095
    hip.set_option("host_direct", 2)
    hip.check(L.dcp_host_register(out.ctypes.data, half, -1))
    try:
        # the kernels no longer store through the half-length alias; the staged path's copy into such a buffer is the runtime's
        # business -- ROCm 7's hipMemcpyAsync refuses a destination that is registered in part ("invalid argument"), which
        # surfaces as a clean error, not a GPU fault
        try:
            res = pp.unwarp_image_backward(img, xc, yc, fact, out=out)
            assert res is out and np.array_equal(out, want)
        except F.HipError as e:
            assert "invalid argument" in str(e)
    finally:
        hip.check(L.dcp_host_unregister(out.ctypes.data))
        hip.set_option("host_direct", 1)
    assert np.array_equal(pp.unwarp_image_backward(img, xc, yc, fact), want)          # the device is healthy afterwards


@pytest.mark.gpu
def test_pinned_output_memory_is_bounded(hip, monkeypatch):
    """results a caller keeps alive are pinned only up to a cap (default: the pool's), and only where the direct path
    can be taken at all."""
    H, W = 2200, 2100                                       # 18.5 MB per result
    img = noise(503, (H, W))
    _pool.clear()
    monkeypatch.setenv("DISCORPY_AMD_PIN_MAX_MB", "40")
    hip.set_option("host_direct", 2)
    try:
        base = _pool.pinned_bytes()
        keep = [pp.unwarp_image_backward(img, 1000.0, 1100.0, [1.0, -8e-6]) for _ in range(4)]
        assert _pool.pinned_bytes() - base <= 40 * (1 << 20) and _pool.pinned_bytes() - base >= H * W * 4      # two pinned, two plain
        assert all(np.array_equal(k, keep[0]) for k in keep)
        hip.set_option("host_direct", 0)                     # the direct path cannot be taken: nothing new is pinned
        now = _pool.pinned_bytes()
        more = pp.unwarp_image_backward(img, 1000.0, 1100.0, [1.0, -8e-6])
        assert _pool.pinned_bytes() == now and np.array_equal(more, keep[0])
        del keep, more
        _pool.clear()
        assert _pool.pinned_bytes() == base
    finally:
        hip.set_option("host_direct", 1)
        _pool.clear()
