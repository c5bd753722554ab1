"""The CPU oracle against the reference's own Fortran, live (oracle/_ref; skipped where that
library is neither buildable nor prebuilt).  Wider sweep than the golden files. Bit-exact."""
import numpy as np
import pytest

import _libs as L
import synth

pytestmark = pytest.mark.skipif(L.ref() is None, reason="reference build (oracle/_ref) not available")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("nx,kind,gd", [(18, "smooth", 8), (35, "checker4", 8), (35, "rough", 8), (35, "smooth", 5)])
def test_fields_rays_bitwise(nx, kind, gd):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    wb = L.RefWB(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    try:
        for a, b in ((wb.gox, g.gox), (wb.goz, g.goz), (wb.dnx, g.dnx), (wb.dnz, g.dnz)):
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)
        pv = synth.medium(nx, kind)
        vr, vo = wb.gridder(pv), L.o_gridder(g, pv)
        assert (bits(vr) != bits(vo)).sum() == 0
        N = g.nnx
        rng = synth.LCG(99 + nx)
        u = rng.uniform(12)
        frac = [(0.43 * N + 0.3, 0.61 * N + 0.6), (1.4, N / 2 + 0.2), (N - 2.5, N - 3.3), (5.0, 7.0)]
        frac += [(u[2 * i] * (N - 1), u[2 * i + 1] * (N - 1)) for i in range(3)]
        for fx, fz in frac:
            sx = np.float32(g.gox + np.float32(fx) * g.dnx)
            sz = np.float32(g.goz + np.float32(fz) * g.dnz)
            r, o = wb.solve(sx, sz), L.o_solve(g, pv, vo, sx, sz)
            assert (bits(r["T"]) != bits(o["T"])).sum() == 0
            assert ((r["Sr"] == 0) != (o["Sr"] == 0)).sum() == 0 and ((r["Sr"] < 0) != (o["Sr"] < 0)).sum() == 0
            live = r["Sr"] >= 0
            assert (bits(r["Tr"])[live] != bits(o["Tr"])[live]).sum() == 0
            v = rng.uniform(8)
            for i in range(4):
                rx = np.float32(g.gox + np.float32(0.2 + v[2 * i] * (N - 1.4)) * g.dnx)
                rz = np.float32(g.goz + np.float32(0.2 + v[2 * i + 1] * (N - 1.4)) * g.dnz)
                assert wb.srtimes(sx, sz, rx, rz).view(np.uint32) == L.o_srtimes(g, vo, o["T"], sx, sz, rx, rz).view(np.uint32)
                fo, _, _ = L.o_rpaths(g, o, vo, sx, sz, rx, rz)
                assert (bits(wb.rpaths(sx, sz, rx, rz)) != bits(fo)).sum() == 0
    finally:
        wb.close()


@pytest.mark.parametrize("nx,kind,period", [(131, "smooth", 3), (131, "rough", 0), (131, "checker", 0)])
def test_fields_bitwise_at_headline_size(nx, kind, period):
    """1025^2 (BASELINE.json configs[2] size): the oracle's field, refined snapshot and receiver times against
    the reference's travel / srtimes (CalSurfG.f90:288-487, :1636-1759), every bit -- the GPU parity tests at
    this size compare with an oracle that is pinned at this size"""
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    wb = L.RefWB(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    try:
        pv = synth.medium(nx, kind, period)
        vr, vo = wb.gridder(pv), L.o_gridder(g, pv)
        assert (bits(vr) != bits(vo)).sum() == 0
        N = g.nnx
        rng = synth.LCG(7 + nx)
        for fx, fz in ((0.37 * (N - 1) + 0.3, 0.58 * (N - 1) + 0.6), (N - 9.25, 300.5)):
            sx = np.float32(g.gox + np.float32(fx) * g.dnx)
            sz = np.float32(g.goz + np.float32(fz) * g.dnz)
            r, o = wb.solve(sx, sz), L.o_solve(g, pv, vo, sx, sz)
            assert (bits(r["T"]) != bits(o["T"])).sum() == 0
            live = r["Sr"] >= 0
            assert (bits(r["Tr"])[live] != bits(o["Tr"])[live]).sum() == 0
            v = rng.uniform(16)
            for i in range(8):
                rx = np.float32(g.gox + np.float32(0.2 + v[2 * i] * (N - 1.4)) * g.dnx)
                rz = np.float32(g.goz + np.float32(0.2 + v[2 * i + 1] * (N - 1.4)) * g.dnz)
                assert wb.srtimes(sx, sz, rx, rz).view(np.uint32) == L.o_srtimes(g, vo, o["T"], sx, sz, rx, rz).view(np.uint32)
    finally:
        wb.close()


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_surfdisp96_bitwise():
    """surfdisp96.f:52-350 on 40 random layered models x (Love, Rayleigh) x (phase, group), spherical"""
    for thk, vpv, vs, rho, t in L.layered_models(40):
        for iwave in (1, 2):
            for igr in (0, 1):
                o = L.surfdisp96("oracle", thk, vpv, vs, rho, 1, iwave, 1, igr, t)
                r = L.surfdisp96("ref", thk, vpv, vs, rho, 1, iwave, 1, igr, t)
                assert (bits64(o) != bits64(r)).sum() == 0
    thk, vpv, vs, rho, t = L.layered_models(1, seed=9)[0]       # flat-earth branch
    for iwave in (1, 2):
        assert (bits64(L.surfdisp96("oracle", thk, vpv, vs, rho, 0, iwave, 1, 0, t)) !=
                bits64(L.surfdisp96("ref", thk, vpv, vs, rho, 0, iwave, 1, 0, t))).sum() == 0


@pytest.mark.parametrize("iwave,igr", [(2, 0), (2, 1), (1, 0), (1, 1)])
def test_depth_kernels_bitwise(iwave, igr):
    """caldespersion (CalSurfG.f90:2320-2368) and depthkernel (:1461-1597) on a 6x5x7 model"""
    c = synth.boundary_case(nx=6, ny=5, nz=7)
    vel = np.ascontiguousarray(c["vels"].T)
    t = np.array([1.0, 2.0, 4.0, 7.0, 11.0, 16.0])
    po = L.depthkernel("oracle", vel, c["depz"], 1.0, iwave, igr, t, kernels=False)
    pr = L.depthkernel("ref", vel, c["depz"], 1.0, iwave, igr, t, kernels=False)
    assert (bits64(po) != bits64(pr)).sum() == 0
    for a, b in zip(L.depthkernel("oracle", vel, c["depz"], 1.0, iwave, igr, t), L.depthkernel("ref", vel, c["depz"], 1.0, iwave, igr, t)):
        assert (bits64(a) != bits64(b)).sum() == 0


@pytest.mark.parametrize("kw", [dict(), dict(deep=True, nz=7, nx=10, ny=13, seed=5), dict(kRc=0, kRg=2, kLc=0, kLg=1)])
def test_boundary_bitwise(kw):
    """The whole boundary: CalSurfG (:939-1459) -> dsurf + CSR-ish rows (rw, iw, col, nar) and
    synthetic (:2412-2865, noiselevel 0) -> obst; all four wave types, ragged source/receiver
    counts, the pvRc overwrite by the group-velocity periods"""
    c = synth.boundary_case(**kw)
    a = L.call_boundary(L.ref().calsurfg_, c)
    b = L.call_boundary(L.oracle().dso_calsurfg, c)
    assert a["nar"] == b["nar"] and a["nar"] > 0
    assert (bits(a["dsurf"]) != bits(b["dsurf"])).sum() == 0
    assert (bits(a["rw"]) != bits(b["rw"])).sum() == 0
    assert (a["iw"] != b["iw"]).sum() == 0 and (a["col"] != b["col"]).sum() == 0
    sa = L.call_boundary(L.ref().synthetic_, c, synthetic=True)
    sb = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)
    assert (bits(sa) != bits(sb)).sum() == 0


def test_aprod_bitwise():
    import ctypes as C
    r = synth.LCG(2)
    m, n, nar = 50, 40, 900
    row = (1 + (r.uniform(nar) * m).astype(np.int32)).clip(1, m).astype(np.int32)
    col = (1 + (r.uniform(nar) * n).astype(np.int32)).clip(1, n).astype(np.int32)
    rw = (r.uniform(nar) - 0.5).astype(np.float32)
    iw = np.concatenate([[nar], row, col]).astype(np.int32)
    ib = lambda v: C.byref(C.c_int(int(v)))
    for mode in (1, 2):
        out = []
        for fn in (L.oracle().dso_aprod, L.ref().aprod_):
            x = (r.uniform(n) if False else np.linspace(-1, 1, n)).astype(np.float32)
            y = np.linspace(2, -2, m).astype(np.float32)
            fn(ib(mode), ib(m), ib(n), L.ptr(x), L.ptr(y), ib(iw.size), ib(nar), L.ptr(iw), L.ptr(rw))
            out.append((x.copy(), y.copy()))
        assert (bits(out[0][0]) != bits(out[1][0])).sum() == 0 and (bits(out[0][1]) != bits(out[1][1])).sum() == 0
