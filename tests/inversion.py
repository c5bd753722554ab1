"""Helpers for the inversion-step tests: the regularised system of main.f90:361-466 built by
the oracle, and one calling convention for the three LSMR implementations (reference module procedure through
oracle/ref_whitebox_lsmr.f90, oracle `dso_lsmr`, product `dsa_lsmr`): every argument by reference like
lsmrModule.f90:36, minus nout."""
import ctypes as C

import numpy as np

import _libs as L


def build_system(c, fwd, obst, threshold0, weight0):
    """fwd: output of L.call_boundary(<calsurfg>, c).  Returns the system LSMR solves as a dict:
    m, n, nar, iw (2 nar + 1: [nar, rows, cols]), rw, b (m), datweight, norm, dws."""
    O = L.oracle()
    nx, ny, nz, dall = c["nx"], c["ny"], c["nz"], c["ndata"]
    maxvp = (nx - 2) * (ny - 2) * (nz - 1)
    cap = fwd["nar"] + 7 * maxvp
    rw = np.zeros(cap, np.float32); rw[:fwd["nar"]] = fwd["rw"]
    col = np.zeros(cap, np.int32); col[:fwd["nar"]] = fwd["col"]
    iw = np.zeros(2 * cap + 1, np.int32); iw[1:fwd["nar"] + 1] = fwd["iw"]
    cbst = np.zeros(dall + maxvp, np.float32)
    datweight = np.zeros(dall, np.float32)
    norm = np.zeros(maxvp, np.float32)
    dws = np.zeros(2, np.float32)
    m, nar = C.c_int(0), C.c_int(0)
    O.dso_iteration_system.argtypes = [C.c_int] * 5 + [C.c_void_p] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 6
    O.dso_iteration_system.restype = None
    O.dso_iteration_system(nx, ny, nz, dall, fwd["nar"], L.ptr(rw), L.ptr(iw), L.ptr(col), L.ptr(np.ascontiguousarray(obst, np.float32)),
                           L.ptr(np.ascontiguousarray(fwd["dsurf"], np.float32)), threshold0, weight0, L.ptr(cbst), L.ptr(datweight),
                           L.ptr(norm), C.byref(m), C.byref(nar), L.ptr(dws))
    n = nar.value
    return dict(m=m.value, n=maxvp, nar=n, iw=iw[:2 * n + 1].copy(), rw=rw[:n].copy(), b=cbst[:m.value].copy(),
                datweight=datweight, norm=norm, dws=dws)


def call_lsmr(fn, S, damp, atol=1e-6, btol=1e-6, conlim=100.0, itnlim=400, local_size=10, head=(), nout=None):
    """main.f90:470-489's call (nout: pass a unit number for entries that keep the reference's full list).  Returns dict(x, istop, itn, normA, condA, normr, normAr, normx)."""
    i32 = lambda v: C.byref(C.c_int(int(v)))
    f32 = lambda v: C.byref(C.c_float(float(v)))
    x = np.zeros(S["n"], np.float32)
    istop, itn = C.c_int(-1), C.c_int(-1)
    sc = [C.c_float(0.0) for _ in range(5)]
    iw = np.ascontiguousarray(S["iw"], np.int32); rw = np.ascontiguousarray(S["rw"], np.float32); b = np.ascontiguousarray(S["b"], np.float32)
    rc = fn(*head, i32(S["m"]), i32(S["n"]), i32(iw.size), i32(rw.size), L.ptr(iw), L.ptr(rw), L.ptr(b), f32(damp), f32(atol), f32(btol),
            f32(conlim), i32(itnlim), i32(local_size), *([] if nout is None else [i32(nout)]), L.ptr(x), C.byref(istop), C.byref(itn), *[C.byref(v) for v in sc])
    if getattr(fn, "__name__", "").startswith("dsa_") and rc != 0:
        raise RuntimeError("%s returned %d" % (fn.__name__, rc))
    return dict(x=x, istop=istop.value, itn=itn.value, normA=np.float32(sc[0].value), condA=np.float32(sc[1].value),
                normr=np.float32(sc[2].value), normAr=np.float32(sc[3].value), normx=np.float32(sc[4].value))


def same(a, b):
    """bitwise comparison of two call_lsmr results; returns the list of differing fields"""
    bad = []
    for k in ("istop", "itn"):
        if a[k] != b[k]:
            bad.append(k)
    for k in ("normA", "condA", "normr", "normAr", "normx"):
        if np.float32(a[k]).view(np.uint32) != np.float32(b[k]).view(np.uint32):
            bad.append(k)
    if (a["x"].view(np.uint32) != b["x"].view(np.uint32)).any():
        bad.append("x")
    return bad
