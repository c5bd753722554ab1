"""The CPU oracle against the committed golden vectors (tests/golden/*.npz, produced by the
reference's own Fortran build -- see tests/golden/make_golden.py).  Bit-exact."""
import glob
import os

import numpy as np
import pytest

import _libs as L

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = sorted(glob.glob(os.path.join(GDIR, "g*.npz")))


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


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_surfdisp96_golden():
    z = np.load(os.path.join(GDIR, "b_disp.npz"))
    for m in range(12):
        for iwave in (1, 2):
            for igr in (0, 1):
                o = L.surfdisp96("oracle", z["thk%d" % m], z["vp%d" % m], z["vs%d" % m], z["rho%d" % m], 1, iwave, 1, igr, z["t%d" % m])
                assert (bits64(o) != bits64(z["c%d_%d%d" % (m, iwave, igr)])).sum() == 0


def test_boundary_golden():
    """one whole CalSurfG + synthetic call and one depthkernel call against the reference's outputs"""
    import synth
    z = np.load(os.path.join(GDIR, "b_boundary.npz"))
    c = synth.boundary_case()
    b = L.call_boundary(L.oracle().dso_calsurfg, c)
    assert b["nar"] == int(z["nar"])
    assert (bits(b["dsurf"]) != bits(z["dsurf"])).sum() == 0 and (bits(b["rw"]) != bits(z["rw"])).sum() == 0
    assert (b["iw"] != z["iw"]).sum() == 0 and (b["col"] != z["col"]).sum() == 0
    assert (bits(L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)) != bits(z["obst"])).sum() == 0
    vel = np.ascontiguousarray(c["vels"].T)
    for a, k in zip(L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), 2, 1, c["tRg"]), ("pvRg", "sen_vsRg", "sen_vpRg", "sen_rhoRg")):
        assert (bits64(a) != bits64(z[k])).sum() == 0


def test_taipei_golden():
    """the oracle on the reference's Taipei example against the reference's own output"""
    from dsurftomo_amd import io as taipei
    z = np.load(os.path.join(GDIR, "b_taipei.npz"))
    c = taipei.load()
    assert c["ndata"] == z["dsurf"].size == 2061 and c["kmax"] == 26
    b = L.call_boundary(L.oracle().dso_calsurfg, c)
    assert b["nar"] == int(z["nar"])
    assert (bits(b["dsurf"]) != bits(z["dsurf"])).sum() == 0
    G = np.zeros((c["ndata"], c["nparpi"]), np.float32)
    G[b["iw"] - 1, b["col"] - 1] = b["rw"]
    assert (G.sum(axis=1, dtype=np.float64) == z["row_sums"]).all()
    assert (np.abs(G).sum(axis=0, dtype=np.float64) == z["col_abs_sums"]).all()
    assert ((G != 0).sum(axis=1) == z["row_counts"]).all()
