"""SURVEY.md section 8(f) rows built so far: f3 calibration-file helpers (CPU) and f1 the colour /
padded unwarp (GPU, marked).  The reference tests restated: tests/test_loadersaver.py:255-291 and
tests/test_utility.py:74-143."""
import os

import numpy as np
import pytest

from conftest import golden, noise

from discorpy_amd.losa import loadersaver as losa
from discorpy_amd.util import utility as util


# ---------------------------------------------------------------- f3: metadata files (CPU)

def test_save_metadata_txt(tmp_path):
    file_path = str(tmp_path / "data" / "coef.txt")
    losa.save_metadata_txt(file_path, 31, 32, (1.0, 0.0))
    assert os.path.isfile(file_path)
    path = losa.save_metadata_txt(file_path, 31, 32, (1.0, 0.0), overwrite=False)
    assert str(path) != file_path and str(path).endswith("coef_0000.txt")
    path2 = losa.save_metadata_txt(file_path, 31, 32, (1.0, 0.0), overwrite=False)
    assert str(path2).endswith("coef_0001.txt")
    losa.save_metadata_txt(str(tmp_path / "data" / "coef1"), 31, 32, (1.0, 0.0))
    assert os.path.isfile(str(tmp_path / "data" / "coef1.txt"))
    assert open(file_path).read() == "xcenter = 31\nycenter = 32\nfactor0 = 1.0\nfactor1 = 0.0\n"


def test_load_metadata_txt_both_separators(tmp_path):
    file_path = str(tmp_path / "coef1.txt")
    losa.save_metadata_txt(file_path, 31.0, 32.0, [1.0, 0.0])
    x, y, facts = losa.load_metadata_txt(file_path)
    assert x == 31.0 and y == 32.0 and facts == [1.0, 0.0]
    # the 'key : value' form of the reference's data/coef_dot_05.txt
    p = tmp_path / "coef_dot_05.txt"
    p.write_text("xcenter : 588.692801577\nycenter : 462.092631791\nfactor0 : 1.00227490554\n"
                 "factor1 : -2.99523692178e-05\nfactor2 : 8.99519088e-08\nfactor3 : -1.57066461911e-10\n"
                 "factor4 : 8.08880211618e-14\n")
    from discorpy_amd import configs
    x, y, facts = losa.load_metadata_txt(str(p))
    assert (x, y, tuple(facts)) == (configs.XCENTER_DOT_05, configs.YCENTER_DOT_05, configs.COEF_DOT_05)
    with pytest.raises(ValueError, match="No such file"):
        losa.load_metadata_txt(str(tmp_path / "missing.txt"))


def test_metadata_json_round_trip(tmp_path):
    file_path = str(tmp_path / "data" / "coef.json")
    losa.save_metadata_json(file_path, 31, 32, [1.0, 0.0])
    assert os.path.isfile(file_path)
    assert str(losa.save_metadata_json(file_path, 31, 32, [1.0, 0.0], overwrite=False)) != file_path
    losa.save_metadata_json(str(tmp_path / "data" / "coef1"), np.float32(31.0), np.int64(32), np.array([1.0, 0.0]))
    x, y, facts = losa.load_metadata_json(str(tmp_path / "data" / "coef1.json"))
    assert x == 31.0 and y == 32.0 and facts == [1.0, 0.0]


# ---------------------------------------------------------------- f1: host-side logic (CPU)

def test_find_point_to_point_closed_form():
    x, y = util.find_point_to_point((10.0, 20.0), 30.0, 25.0, [1.0, 1e-2])
    r = np.hypot(20.0 - 30.0, 10.0 - 25.0)
    assert np.isclose(x, 30.0 + (1 + 1e-2 * r) * (20.0 - 30.0)) and np.isclose(y, 25.0 + (1 + 1e-2 * r) * (10.0 - 25.0))
    assert util.find_point_to_point((10.0, 20.0), 30.0, 25.0, [1.0, 1e-2], output_order="yx") == (y, x)


def test_pad_argument_validation():
    args = (64, 64, 32.0, 32.0, [1.0, 1e-3])
    assert util._calc_pad(False, *args) == (0, 0, 0, 0)
    assert util._calc_pad(5, *args) == (5, 5, 5, 5)
    assert util._calc_pad((1, 2, 3, 4), *args) == (1, 2, 3, 4)
    with pytest.raises(ValueError, match="Incorrect format"):
        util._calc_pad((1, 2, 3), *args)
    with pytest.raises(ValueError, match="Invalid format"):
        util._calc_pad("2", *args)



def test_g14_model_reversal_and_automatic_pads():
    """proc.transform_coef_backward_and_forward restated (processing.py:615-674) and the pad=True widths of
    utility.py:238-263, against the reference's own numbers (golden G14)."""
    g = golden("g14_autopad40x56x3")
    fact = list(g["list_fact"])
    grid = [[gy - 19.1, gx - 27.4] for gy in np.linspace(0, 40, 40) for gx in np.linspace(0, 56, 40)]
    assert np.allclose(util.transform_coef_backward_and_forward(fact, ref_points=grid), g["forward_fact"], rtol=1e-9, atol=0)
    assert np.allclose(util.transform_coef_backward_and_forward([1.0, -3e-5, 9e-8]), g["default_grid_backward"], rtol=1e-9, atol=0)
    assert np.allclose(util.transform_coef_backward_and_forward([1.0, -3e-5, 9e-8], mapping="forward"),
                       g["default_grid_forward"], rtol=1e-9, atol=0)
    assert util._calc_pad(True, 40, 56, 27.4, 19.1, fact) == tuple(int(v) for v in g["pads"]) == (8, 7, 11, 11)
    # the reference's own pad=True case (tests/test_utility.py:92-101)
    ref = [[i - 20.0, j - 30.0] for i in range(0, 40, 10) for j in range(0, 60, 10)]
    tfact = util.transform_coef_backward_and_forward([1.0, 0.1, 0.01], ref_points=ref)
    assert np.allclose(tfact, g["reftest_tfact"], rtol=1e-9, atol=0)
    assert util._calc_pad(True, 40, 60, 30.0, 20.0, tfact) == tuple(int(v) for v in g["reftest_pads"])
    with pytest.raises(ValueError, match="Number of reference-points"):
        util.transform_coef_backward_and_forward([1.0, 0.1, 0.01], ref_points=[[1.0, 2.0]])


# ---------------------------------------------------------------- f1: the kernels (GPU)

@pytest.mark.gpu
def test_g10_color_unwarp_matches_the_reference(hip):
    g = golden("g10_color40x56x3")
    rgb = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    out = util.unwarp_color_image_backward(rgb, *a, blend="scipy")
    assert out.shape == (40, 56, 3) and np.array_equal(out, g["nopad"])
    out = util.unwarp_color_image_backward(rgb, *a, pad=(3, 5, 2, 7), pad_mode="edge", blend="scipy")
    assert out.shape == (48, 65, 3) and np.array_equal(out, g["pad_3_5_2_7_edge"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb, *a, pad=4, blend="scipy"), g["pad_4_constant"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb, *a, order=0, pad=4, pad_mode="reflect"),
                          g["pad_4_reflect_order0"])
    gray = util.unwarp_color_image_backward(rgb[:, :, 1], *a, pad=6, pad_mode="mean", blend="scipy")
    assert gray.shape == (52, 68) and np.array_equal(gray, g["gray_pad_6_mean"])
    # the reference's own shape / mean assertions (tests/test_utility.py:74-143)
    assert util.unwarp_color_image_backward(rgb, *a, pad=(1, 2, 3, 4)).shape == (43, 63, 3)
    assert np.mean(util.unwarp_color_image_backward(rgb, *a)) < 1.0


@pytest.mark.gpu
def test_color_unwarp_on_device_tensors(hip):
    torch = pytest.importorskip("torch")
    g = golden("g10_color40x56x3")
    rgb = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    t = torch.from_numpy(rgb).cuda()
    out = util.unwarp_color_image_backward(t, *a, pad=(3, 5, 2, 7), pad_mode="edge", blend="scipy")
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), g["pad_3_5_2_7_edge"])
    assert np.array_equal(util.unwarp_color_image_backward(t, *a, pad=4, blend="scipy").cpu().numpy(), g["pad_4_constant"])
    assert np.array_equal(util.unwarp_color_image_backward(t, *a, order=0, pad=4, pad_mode="reflect").cpu().numpy(),
                          g["pad_4_reflect_order0"])
    with pytest.raises(NotImplementedError, match="pad_mode"):
        util.unwarp_color_image_backward(t, *a, pad=4, pad_mode="mean")


@pytest.mark.gpu
def test_g14_pad_true_matches_the_reference(hip):
    """pad=True end to end (the call form of docs/source/technical_notes/fisheye_correction.rst:374) without the
    reference installed: pad widths from the restated fit, padded unwarp on the GPU, bit-equal to golden G14."""
    g = golden("g14_autopad40x56x3")
    rgb = noise(g["seed"], g["shape"])
    a = (float(g["xcenter"]), float(g["ycenter"]), list(g["list_fact"]))
    out = util.unwarp_color_image_backward(rgb, *a, pad=True, blend="scipy")
    assert out.shape == (40 + 8 + 7, 56 + 11 + 11, 3) and np.array_equal(out, g["pad_true_constant"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb, *a, order=0, pad=True, pad_mode="edge"), g["pad_true_edge_order0"])
    assert np.array_equal(util.unwarp_color_image_backward(rgb[:, :, 2], *a, pad=True, pad_mode="reflect", blend="scipy"),
                          g["gray_pad_true_reflect"])
    # tests/test_utility.py:75-101 restated
    box = np.ones((40, 60), dtype=np.float32)
    box[:5] = 0
    box[-5:] = 0
    box[:, :5] = 0
    box[:, -5:] = 0
    ref = [[i - 20.0, j - 30.0] for i in range(0, 40, 10) for j in range(0, 60, 10)]
    tfact = util.transform_coef_backward_and_forward([1.0, 0.1, 0.01], ref_points=ref)
    cor = util.unwarp_color_image_backward(box, 30.0, 20.0, tfact, 1, "constant", True, "constant", blend="scipy")
    assert cor.shape != (40, 60) and cor.shape == g["reftest_pad_true"].shape
    assert np.max(np.abs(cor - g["reftest_pad_true"])) <= 1e-5    # tfact from lstsq may differ in its last bits
