"""8- / 16-bit integer data: tiles whose source box lies at coordinates >= 32 blend in the factorised form, which is exact there
(every product and sum fits 53 bits: unwarp_kernels.hip, exact_lerp_pairs) and therefore equal to scipy's operation order bit for
bit.  Checked against the oracle (scipy's order everywhere) and against the same kernels with the option switched off.
Reference behaviour: output dtype = input dtype, scipy's integer rounding (postprocessing.py:147, 251)."""
import numpy as np
import pytest

from conftest import HOST, oblend, typed_image

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
        F.set_option("x_int_exact", mode)
        try:
            got[mode] = (pp.unwarp_image_backward(img, xc, yc, fact), pp.correct_perspective_image(img, coef),
                         pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1))
            kernel = F.last_kernel()
        finally:
            F.set_option("x_int_exact", 1)
    assert "stack_wg_kernel" in kernel or "stack" in kernel
    for a, b in zip(got[0], got[1]):
        assert a.dtype == np.dtype(dt) and np.array_equal(a, b)
    assert np.array_equal(got[1][0], orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL))
    assert np.array_equal(got[1][1], orc.correct_perspective_image(img, coef))
    assert np.array_equal(got[1][2], orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1, poly=orc.POLY_KERNEL))


@pytest.mark.parametrize("nterms", [1, 2, 3])
def test_short_coefficient_vectors_run_the_four_term_kernels_zero_padded(hip, orc, nterms):
    """Fewer than four coefficients on integer data: the NF = 4 instantiations on a zero-padded vector (fma(r2, 0, a) = a exactly),
    not the run-time-length ones -- frames, stacks and centre grids, all equal to the oracle."""
    H, W = 300, 700
    fact = [1.002, -3e-5, 5e-8][:nterms]
    xc, yc = 330.4, 151.2
    assert F.tile_certificate(H, W, xc, yc, fact) == 2
    img = typed_image("uint16", (H, W), 77)
    got = pp.unwarp_image_backward(img, xc, yc, fact)
    assert F.last_kernel().startswith("remap_wg_kernel<Radial,NF=4,"), F.last_kernel()
    assert np.array_equal(got, orc.unwarp_image_backward(img, xc, yc, fact, poly=orc.POLY_KERNEL))
    vol = typed_image("uint16", (5, H, W), 78)
    F.set_option("x_stack_wg", 2)
    try:
        got = pp.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1)
        assert "NF=4" in F.last_kernel(), F.last_kernel()
    finally:
        F.set_option("x_stack_wg", 1)
    assert np.array_equal(got, orc.unwarp_chunk_slices_backward(vol, xc, yc, fact, 0, H - 1, poly=orc.POLY_KERNEL))
    volf = np.random.default_rng(5).random((4, H, W), dtype=np.float32)
    cents = [(xc + 3.5 * i, yc - 2.25 * i) for i in range(3)]
    got = pp.unwarp_slice_backward_centres(volf, [c[0] for c in cents], [c[1] for c in cents], fact, 140)
    assert "NF=4" in F.last_kernel(), F.last_kernel()
    for k, (cx, cy) in enumerate(cents):
        assert np.array_equal(got[k], orc.unwarp_slice_backward(volf, cx, cy, fact, 140, poly=orc.POLY_KERNEL, blend=oblend(orc, HOST)))
