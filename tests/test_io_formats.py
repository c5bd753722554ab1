"""Readers and writers of the reference's file formats (dsurftomo_amd/io.py, invert.py): no GPU involved."""
import os

import numpy as np

from dsurftomo_amd import invert, io


def test_taipei_input_is_read_like_the_reference():
    """DSurfTomo.in / surfdataTB.dat / MOD of the reference's example (main.f90:134-335)"""
    c = io.load()
    assert (c["nx"], c["ny"], c["nz"]) == (18, 18, 9) and c["kRc"] == 26 and c["kRg"] == c["kLc"] == c["kLg"] == 0
    assert c["ndata"] == 2061 and c["nparpi"] == 16 * 16 * 8 and c["nsrcsurf"] == 20
    assert float(c["weight0"]) == 4.0 and float(c["damp"]) == 1.0 and float(c["threshold0"]) == 3.0
    assert float(c["minvel"]) == 0.5 and abs(float(c["maxvel"]) - 2.8) < 1e-6 and c["maxiter"] == 10 and c["ifsyn"] == 0
    assert abs(c["spfra"] - 0.2) < 1e-12 and abs(float(c["noiselevel"]) - 0.02) < 1e-7
    assert np.allclose(c["tRc"], np.arange(5, 31) / 10.0)
    assert c["vels"].shape == (18, 18, 9) and c["vels"].flags.f_contiguous and c["depz"].shape == (9,)
    # observed times = distance / velocity, one per datum, all positive; sources in colatitude / longitude radians
    assert c["obst"].shape == (2061,) and (c["obst"] > 0).all() and (c["dist"] > 0).all()
    assert int(c["nrc1"].sum()) == 2061 and (c["nsrcsurf1"] <= 20).all()
    k = 0
    s = c["scxf"][:c["nsrcsurf1"][k], k]
    assert ((s > 1.1) & (s < 1.2)).all()                                   # colatitude of ~24.9..25.2 degrees north


def test_model_writer_format(tmp_path):
    """'(5f10.5)' lines in k / j / i order: longitude, latitude, depth, Vs (main.f90:537-546)"""
    c = io.load()
    vs = np.asfortranarray(c["vels"].copy())
    vs[1, 1, 0] = np.float32(1.23456789)
    path = str(tmp_path / "model.dat")
    invert.write_model(path, c, vs)
    lines = open(path).read().splitlines()
    assert len(lines) == 16 * 16 * 8 and all(len(l) == 40 for l in lines)
    assert lines[0] == " 121.35000  25.20000   0.00000   1.23457"
    a = np.loadtxt(path)
    assert np.allclose(a[1, :2], [121.35, 25.2 - 0.015], atol=1e-5)         # i runs fastest: latitude decreases
    assert np.allclose(a[16, :2], [121.35 + 0.017, 25.2], atol=1e-5)        # then j: longitude increases
    assert abs(a[256, 2] - c["depz"][1]) < 1e-5
    assert np.abs(a[:, 3].reshape(8, 16, 16).transpose(2, 1, 0) - vs[1:-1, 1:-1, :-1]).max() < 6e-6


def test_residual_writer(tmp_path):
    c = io.load()
    d = c["obst"] * np.float32(0.9)
    w = (np.arange(c["ndata"]) % 3 != 0).astype(np.float32)
    path = str(tmp_path / "residualFirst.dat")
    invert.write_residuals(path, c, d, c["obst"], w)
    a = np.loadtxt(path)
    assert a.shape == (c["ndata"], 6)
    assert np.allclose(a[:, 0], c["dist"], rtol=1e-6) and np.allclose(a[:, 3], d * w, rtol=1e-6) and (a[:, 5] == w).all()
