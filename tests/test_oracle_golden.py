"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, produced by the
reference's own Fortran build -- see tests/golden/make_golden.py).  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import _libs as L

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_golden_files_exist():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_golden(path):
    z = np.load(path)
    nx, gd = int(z["nx"]), int(z["gd"])
    g = L.grid(nx, nx, float(z["goxd"]), float(z["gozd"]), float(z["dvd"]), float(z["dvd"]), gd)
    pv = z["pv"]
    veln = L.o_gridder(g, pv)
    assert (bits(veln) != bits(z["veln"])).sum() == 0
    for k, (sx, sz) in enumerate(z["src"]):
        o = L.o_solve(g, pv, veln, sx, sz)
        assert (bits(o["T"]) != bits(z["T%d" % k])).sum() == 0
        cls = np.sign(o["Sr"]).clip(-1, 1).astype(np.int8)
        assert (cls != z["Sr%d" % k]).sum() == 0
        live = cls >= 0
        assert (bits(o["Tr"])[live] != bits(z["Tr%d" % k])[live]).sum() == 0
        icls = np.sign(o["inj_s"]).clip(-1, 1).astype(np.int8)
        assert (icls != z["injS%d" % k]).sum() == 0
        m = icls >= 0
        assert (bits(o["inj_t"])[m] != bits(z["injT%d" % k])[m]).sum() == 0
        for r, (rx, rz) in enumerate(z["rec"]):
            t = L.o_srtimes(g, veln, o["T"], sx, sz, rx, rz)
            assert t.view(np.uint32) == z["t%d" % k][r].view(np.uint32)
        for r, (rx, rz) in enumerate(z["rec"][:3]):
            fdm, _, _ = L.o_rpaths(g, o, veln, sx, sz, rx, rz)
            assert (bits(fdm) != bits(z["fdm%d" % k][r])).sum() == 0
