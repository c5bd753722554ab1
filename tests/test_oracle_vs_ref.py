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
