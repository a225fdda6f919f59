"""SURVEY.md section 5's hooks: the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer, and the bounds-checking build
of the kernels (every LDS tap of the staged kernels checked against its slab).  The sanitizer run is CPU-only; the bounds-checking
library is exercised by tools/fuzz_parity.py --bounds (profiles/rounds_1-4/r04*_fuzz_bounds_build.txt) and, when it has been built, here."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, noise


def test_oracle_golden_suite_under_asan_and_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    r = subprocess.run([os.path.join(ROOT, "tools", "oracle_asan.sh"), "-k", "g1_ or g2_ or g5 or g8 or g9 or g11 or g12 or g15 or g17"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.gpu
def test_product_build_compiles_no_bounds_checks_and_the_checking_build_counts_none(hip):
    assert hip.debug_bounds() == (0, 0, 0, 0, 0)                    # the product library: no check compiled in, says so
    lib = os.path.join(ROOT, "discorpy_amd", "lib", "libdiscorpy_hip_bounds.so")
    if not os.path.exists(lib):
        pytest.skip("make -C discorpy_amd/csrc bounds has not been run in this tree")
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import noise
from discorpy_amd import _ffi as F
from discorpy_amd.post import postprocessing as pp
from discorpy_amd.util import utility as util
F.require_device()
assert F.debug_bounds()[4] == 1
img = noise(5, (700, 1100))
fact = [1.0, -2e-5, 3e-8]
pp.unwarp_image_backward(img, 500.0, 333.0, fact); k1 = F.last_kernel()
pp.unwarp_image_backward(img, 500.0, 333.0, fact, order=3)
util.unwarp_color_image_backward(noise(6, (300, 500, 3)), 250.0, 140.0, fact); k2 = F.last_kernel()
pp.unwarp_chunk_slices_backward(noise(7, (6, 300, 520)), 250.0, 160.0, fact, 20, 90)
b = F.debug_bounds()
assert k1.startswith("remap_wg_kernel") and k2.startswith("remap_wg_color_kernel"), (k1, k2)
assert b[0] == 0 and b[4] == 1, b
print("bounds ok", b)
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, DCP_LIB_PATH=lib))
    assert r.returncode == 0 and "bounds ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
