"""Host glue of one outer iteration in the library (csrc/iteration.hip: dsa_iteration_system, dsa_model_update) against
the oracle's restatement of main.f90:361-466 / :520-535.  Plain host code: runs without a GPU."""
import ctypes as C

import numpy as np

import _libs as L
import inversion as inv
import synth
from dsurftomo_amd.engine import load_library


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def product_system(c, fwd, obst, threshold0, weight0, slack=0):
    lib = load_library()
    nx, ny, nz, dall = c["nx"], c["ny"], c["nz"], c["ndata"]
    maxvp = (nx - 2) * (ny - 2) * (nz - 1)
    cap = fwd["nar"] + 7 * maxvp + slack
    rw = np.zeros(cap, np.float32); rw[:fwd["nar"]] = fwd["rw"]
    col = np.zeros(cap, np.int32); col[:fwd["nar"]] = fwd["col"]
    iw = np.zeros(2 * cap + 1, np.int32); iw[1:fwd["nar"] + 1] = fwd["iw"]
    cbst = np.zeros(dall + maxvp, np.float32); datweight = np.zeros(dall, np.float32)
    norm = np.zeros(maxvp, np.float32); dws = np.zeros(2, np.float32)
    m, nar = C.c_int(0), C.c_longlong(0)
    lib.dsa_iteration_system.argtypes = [C.c_int] * 4 + [C.c_longlong] * 2 + [C.c_void_p] * 5 + [C.c_float] * 2 + [C.c_void_p] * 6
    rc = lib.dsa_iteration_system(nx, ny, nz, dall, fwd["nar"], cap, L.ptr(rw), L.ptr(iw), L.ptr(col), L.ptr(np.ascontiguousarray(obst, np.float32)),
                                  L.ptr(np.ascontiguousarray(fwd["dsurf"], np.float32)), threshold0, weight0, L.ptr(cbst), L.ptr(datweight),
                                  L.ptr(norm), C.byref(m), C.byref(nar), L.ptr(dws))
    assert rc == 0
    n = nar.value
    return dict(m=m.value, n=maxvp, nar=n, iw=iw[:2 * n + 1].copy(), rw=rw[:n].copy(), b=cbst[:m.value].copy(), datweight=datweight, norm=norm, dws=dws)


def test_iteration_system_matches_oracle():
    for seed, (thr, w) in enumerate(((3.0, 2.0), (1.0, 0.5), (10.0, 4.0))):
        c = synth.boundary_case(seed=synth.SEED + seed)
        fwd = L.call_boundary(L.oracle().dso_calsurfg, c)
        r = synth.LCG(5 + seed)
        obst = (fwd["dsurf"] * (1.0 + 0.1 * (r.uniform(c["ndata"]) - 0.4))).astype(np.float32)
        a = inv.build_system(c, fwd, obst, thr, w)
        b = product_system(c, fwd, obst, thr, w)
        assert a["m"] == b["m"] and a["nar"] == b["nar"] and (a["iw"] == b["iw"]).all()
        for k in ("rw", "b", "datweight", "norm", "dws"):
            assert (bits(a[k]) == bits(b[k])).all(), k
        assert 0 < (b["datweight"] == 0).sum() < c["ndata"] or thr == 10.0


def test_capacity_is_checked():
    c = synth.boundary_case()
    fwd = L.call_boundary(L.oracle().dso_calsurfg, c)
    lib = load_library()
    maxvp = c["nparpi"]
    cap = fwd["nar"] + 10
    rw = np.zeros(cap, np.float32); col = np.zeros(cap, np.int32); iw = np.zeros(2 * cap + 1, np.int32)
    cbst = np.zeros(c["ndata"] + maxvp, np.float32); dw = np.zeros(c["ndata"], np.float32); norm = np.zeros(maxvp, np.float32); dws = np.zeros(2, np.float32)
    m, nar = C.c_int(0), C.c_longlong(0)
    lib.dsa_iteration_system.argtypes = [C.c_int] * 4 + [C.c_longlong] * 2 + [C.c_void_p] * 5 + [C.c_float] * 2 + [C.c_void_p] * 6
    rc = lib.dsa_iteration_system(c["nx"], c["ny"], c["nz"], c["ndata"], fwd["nar"], cap, L.ptr(rw), L.ptr(iw), L.ptr(col), L.ptr(fwd["dsurf"]), L.ptr(fwd["dsurf"]),
                                  3.0, 1.0, L.ptr(cbst), L.ptr(dw), L.ptr(norm), C.byref(m), C.byref(nar), L.ptr(dws))
    assert rc == -6                                                  # DSA_ERR_CAPACITY


def test_model_update_matches_oracle():
    lib, O = load_library(), L.oracle()
    nx, ny, nz = 9, 8, 6
    r = synth.LCG(12)
    dv0 = (1.6 * (r.uniform((nx - 2) * (ny - 2) * (nz - 1)) - 0.5)).astype(np.float32)
    vs0 = (2.0 + 2.5 * r.uniform(nx * ny * nz)).astype(np.float32)
    out = []
    for fn in (lib.dsa_model_update, O.dso_model_update):
        fn.argtypes = [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_float] * 2
        dv, vs = dv0.copy(), vs0.copy()
        fn(nx, ny, nz, L.ptr(dv), L.ptr(vs), 2.3, 4.2)
        out.append((dv, vs))
    assert (bits(out[0][0]) == bits(out[1][0])).all() and (bits(out[0][1]) == bits(out[1][1])).all()
    assert np.abs(out[0][0]).max() == 0.5 and out[0][1].reshape(nz, ny, nx)[:-1, 1:-1, 1:-1].max() <= np.float32(4.2)
    # the boundary columns and the bottom layer stay as they were
    v0, v1 = vs0.reshape(nz, ny, nx), out[0][1].reshape(nz, ny, nx)
    assert (v0[-1] == v1[-1]).all() and (v0[:, 0] == v1[:, 0]).all() and (v0[:, :, -1] == v1[:, :, -1]).all()
