"""8- / 16-bit integer data: tiles whose source box lies at coordinates >= 32 blend in the factorised form, which is exact there
(every product and sum fits 53 bits: unwarp_kernels.hip, exact_lerp_pairs) and therefore equal to scipy's operation order bit for
bit.  Checked against the oracle (scipy's order everywhere) and against the same kernels with the option switched off.
Reference behaviour: output dtype = input dtype, scipy's integer rounding (postprocessing.py:147, 251)."""
import numpy as np
import pytest

from conftest import typed_image

from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", ["uint16", "int16", "uint8", "int8"])
def test_integer_frames_and_stacks_exact_blend_equals_scipy_order(hip, orc, dt):
    H, W = 420, 1100                                   # tiles at the top / left border (coordinates < 32) and far inside
    img = typed_image(dt, (H, W), 31)
    xc, yc, fact = 540.3, 207.9, [1.001, -4e-5, 6e-8, -2e-11]
    assert F.tile_certificate(H, W, xc, yc, fact) == 2
    coef = [0.99, 0.004, 3.1, -0.003, 1.004, 1.7, 2e-6, -1e-6]
    vol = typed_image(dt, (6, H, W), 32)
    got = {}
    for mode in (1, 0):
        F.set_option("int_exact", mode)
        try:
            got[mode] = (pp.unwarp_image_backward(img, xc, yc, fact), pp.correct_perspective_image(img, coef),
                         pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1))
            kernel = F.last_kernel()
        finally:
            F.set_option("int_exact", 1)
    assert "stack_wg_kernel" in kernel or "stack" in kernel
    for a, b in zip(got[0], got[1]):
        assert a.dtype == np.dtype(dt) and np.array_equal(a, b)
    assert np.array_equal(got[1][0], orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL))
    assert np.array_equal(got[1][1], orc.correct_perspective_image(img, coef))
    assert np.array_equal(got[1][2], orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1, poly=orc.POLY_KERNEL))
