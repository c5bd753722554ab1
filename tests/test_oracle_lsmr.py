"""Oracle side of the inversion step (lsmr_oracle.c) against the reference's own objects (oracle/_ref) and the
committed golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest

import _libs as L
import inversion as inv
import synth

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def small_system(seed=0, damp_rows=True):
    c = synth.boundary_case()
    fwd = L.call_boundary(L.oracle().dso_calsurfg, c)
    r = synth.LCG(77 + seed)
    obst = (fwd["dsurf"] * (1.0 + 0.04 * (r.uniform(c["ndata"]) - 0.5))).astype(np.float32)
    return c, fwd, inv.build_system(c, fwd, obst, 3.0, 2.0)


needs_ref = pytest.mark.skipif(L.ref() is None, reason="reference build not available")


@needs_ref
def test_blas_and_percentile_bitwise_against_reference():
    R, O = L.ref(), L.oracle()
    O.dso_dnrm2.restype = C.c_float; O.dso_dnrm2.argtypes = [C.c_int, C.c_void_p]
    R.dnrm2_.restype = C.c_float
    r = synth.LCG(5)
    for n in (1, 2, 7, 64, 1000, 4097):
        for kind in range(4):
            x = (r.uniform(n) - 0.5).astype(np.float32)
            if kind == 1: x[::3] = 0.0
            if kind == 2: x = np.sort(np.abs(x)).astype(np.float32)                 # every element a new maximum
            if kind == 3: x *= np.float32(1e-20)
            a = np.float32(O.dso_dnrm2(n, L.ptr(x)))
            b = np.float32(R.dnrm2_(C.byref(C.c_int(n)), L.ptr(x), C.byref(C.c_int(1))))
            assert a.view(np.uint32) == b.view(np.uint32), (n, kind)
    O.dso_getpercentile.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in (5, 100, 2061):
        x = (r.uniform(n) - 0.3).astype(np.float32)
        qa, qb = np.zeros(2, np.float32), np.zeros(2, np.float32)
        O.dso_getpercentile(n, L.ptr(x), L.ptr(qa[0:]), L.ptr(qa[1:]))
        R.getpercentile_(C.byref(C.c_int(n)), L.ptr(x), L.ptr(qb[0:]), L.ptr(qb[1:]))
        assert (bits(qa) == bits(qb)).all()


@needs_ref
@pytest.mark.parametrize("damp,local_size,itnlim", [(1.0, 10, 400), (0.0, 10, 60), (0.5, 0, 100), (1.0, 3, 7)])
def test_lsmr_bitwise_against_reference(damp, local_size, itnlim):
    """oracle LSMR == the reference's module procedure on a regularised system from the boundary case"""
    c, fwd, S = small_system()
    a = inv.call_lsmr(L.oracle().dso_lsmr, S, damp, itnlim=itnlim, local_size=local_size)
    b = inv.call_lsmr(L.ref().ref_wb_lsmr, S, damp, itnlim=itnlim, local_size=local_size)
    assert a["itn"] > 3 and np.abs(a["x"]).max() > 0
    assert inv.same(a, b) == []


def test_iteration_system_shape():
    """rows, columns and right-hand side of main.f90:361-466 on the boundary case"""
    c, fwd, S = small_system()
    nvx, nvz, nl = c["nx"] - 2, c["ny"] - 2, c["nz"] - 1
    assert S["m"] == c["ndata"] + nvx * nvz * nl and S["n"] == nvx * nvz * nl
    interior = (nvx - 2) * (nvz - 2) * (nl - 2)
    assert S["nar"] == fwd["nar"] + 7 * interior + (S["n"] - interior)
    assert S["iw"][0] == S["nar"] and (S["b"][c["ndata"]:] == 0).all()
    rows, cols = S["iw"][1:S["nar"] + 1], S["iw"][S["nar"] + 1:]
    assert rows.min() >= 1 and rows.max() == S["m"] and cols.min() >= 1 and cols.max() <= S["n"]
    # data rows were scaled by their weights, DWS is the column sum of |G|
    w = S["datweight"][fwd["iw"] - 1]
    assert (bits(S["rw"][:fwd["nar"]]) == bits(fwd["rw"] * w)).all()
    dws = np.zeros(S["n"], np.float64); np.add.at(dws, fwd["col"] - 1, np.abs(fwd["rw"] * w).astype(np.float64))
    assert np.allclose(S["norm"], dws, rtol=1e-5)


def test_lsmr_golden():
    """the oracle against the vectors the reference produced (tests/golden/make_golden.py)"""
    z = np.load(os.path.join(GDIR, "b_lsmr.npz"))
    c, fwd, S = small_system()
    assert (S["iw"] == z["iw"]).all() and (bits(S["rw"]) == bits(z["rw"])).all() and (bits(S["b"]) == bits(z["b"])).all()
    a = inv.call_lsmr(L.oracle().dso_lsmr, S, 1.0)
    assert a["itn"] == int(z["itn"]) and a["istop"] == int(z["istop"])
    assert (bits(a["x"]) == bits(z["x"])).all()
    for k in ("normA", "condA", "normr", "normAr", "normx"):
        assert np.float32(a[k]).view(np.uint32) == np.float32(z[k]).view(np.uint32)
