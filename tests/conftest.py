import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# blend=None: host (NumPy) inputs get scipy's exact operation order, device-resident ones the factorised float64 form
# (discorpy_amd/post/postprocessing.py _default_blend).  DISCORPY_AMD_HOST_BLEND moves the host default -- a run of the GPU suite
# under DISCORPY_AMD_HOST_BLEND=f32 fails every test that names the wrong default for the kind of array it passes.
HOST = os.environ.get("DISCORPY_AMD_HOST_BLEND", "scipy")
DEV = "f64lerp"


def oblend(orc, name):
    """The oracle's code of a blend name ("scipy", "f64lerp", "f32")."""
    return {"scipy": orc.BLEND_SCIPY, "f64lerp": orc.BLEND_F64LERP, "f32": orc.BLEND_F32LERP}[name]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (built on demand with gcc) -- the checker, never the thing under test."""
    from oracle import oracle as o
    o.build()
    o.set_threads(min(16, o.max_threads()))
    return o


@pytest.fixture(scope="session")
def hip():
    """The HIP library with a visible device; fails (not skips) if either is missing on a GPU run."""
    from discorpy_amd import _ffi as F
    if not os.path.exists(F.LIB_PATH):
        # a fresh checkout on the GPU box: compile in-tree (hipcc is part of the image)
        import __graft_entry__
        __graft_entry__.build()
    F.lib()
    F.require_device()
    return F


def noise(seed, shape):
    return np.random.default_rng(int(seed)).random(tuple(int(v) for v in shape), dtype=np.float32)


def typed_image(dt, shape, seed):
    """The inputs of golden G12 (tools/gen_golden.py): full-range integers / floats in [-700, 1300)."""
    rng = np.random.default_rng(int(seed))
    dt = np.dtype(dt)
    shape = tuple(int(v) for v in shape)
    if dt.kind == "f":
        return (rng.random(shape) * 2000.0 - 700.0).astype(dt)
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=np.int64).astype(dt)


def g12_inputs(g, dt):
    """(image, volume) of golden G12 for element type `dt`."""
    names = ("uint8", "int8", "uint16", "int16", "uint32", "int32", "float64")
    k = names.index(dt)
    im = typed_image(dt, g["shape"], 800 + k)
    im[:8, :5] = np.arange(40).reshape(8, 5).astype(im.dtype) - (20 if np.dtype(dt).kind != "u" else 0)
    return im, typed_image(dt, g["vol_shape"], 900 + k)


def wide_image(dt, shape, seed):
    """The inputs of golden G12b (tools/gen_golden.py): int64 / uint64 over the whole range with the extremes and small values
    sprinkled in, bool coin flips."""
    rng = np.random.default_rng(int(seed))
    shape = tuple(int(v) for v in shape)
    if np.dtype(dt) == np.bool_:
        return rng.random(shape) < 0.5
    info = np.iinfo(dt)
    im = rng.integers(info.min, info.max, size=shape, endpoint=True, dtype=dt)
    flat = im.reshape(-1)
    flat[::7] = rng.integers(0, 1 << 20, size=flat[::7].shape).astype(dt)
    flat[3::11] = info.max
    flat[5::13] = info.min
    return im


G12B_DTYPES = ("int64", "uint64", "bool")
G12_DTYPES = ("uint8", "int8", "uint16", "int16", "uint32", "int32", "float64")


def ulp_diff(a, b):
    """Distance in float32 ulps between two float32 arrays (finite values)."""
    ia = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)
