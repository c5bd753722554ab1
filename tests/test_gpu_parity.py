"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerance (BASELINE.json north_star): travel times within 1e-4 s of the reference FMM on
identical grids.  Integer/status outputs and the diced velocity grids are compared bit-exactly.
"""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import parity_log
import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

# Modes (csrc/engine.h): the DEFAULT is exact_ties = 1 -- fixed point, census of its exact ties, flagged units solved again by the reference's
# march -- and must hold 1e-4 s on every field of this file.  exact_ties = 0 (the fixed point alone) is run beside it: it holds the bar too except
# on the one exact-tie case named here: two neighbouring narrow-band nodes carry bit-equal times, the reference pops one of them first (which
# one is decided by its heap layout, CalSurfG.f90:417-424 / :768-921) and re-evaluates the other against it at the double root of the two-sided
# quadratic; the fixed-point solve accepts both without using either in the other's stencil.  That field is reported and bounded by a band
# (measured: 4.13e-4 s on 41 of 66 049 nodes), not pinned.
TIE_CASES = {
    (35, "checker4", 8, 3): (6e-4, 80),     # source on the node (5, 7) of a +-13 % checkerboard
}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def oracle_case(nx, kind, gd, srcs):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, kind)
    veln = L.o_gridder(g, pv)
    sols = [L.o_solve(g, pv, veln, sx, sz) for sx, sz in srcs]
    return g, pv, veln, sols


def positions(nx, gd, frac):
    gox, goz, dnx, dnz = synth.grid_origin(nx, gd)
    N = synth.nprop(nx, gd)
    out = []
    for fx, fz in frac:
        fx = np.float32(fx * (N - 1) if fx <= 1.0 else fx)
        fz = np.float32(fz * (N - 1) if fz <= 1.0 else fz)
        out.append((np.float32(gox + fx * dnx), np.float32(goz + fz * dnz)))
    return out


CASES = [
    (18, "homog", 8), (18, "smooth", 8), (35, "smooth", 8), (35, "checker4", 8), (35, "smooth", 5), (35, "rough", 8),
]
FRAC = [(0.43, 0.61), (1.4, 0.5), (0.985, 0.99), (5.0, 7.0), (0.93, 3.2), (0.5, 0.5), (16.0, 24.0), (1.0, 1.0), (0.0, 0.0),
        (0.21, 0.77), (0.66, 0.12),
        # found by tests/tools/fuzz_parity.py: 2.5 cells from the high-x edge at N = 121, where the refined stage ends early
        # and the coarse tree starts out of order (a second-order leg must not use a node accepted after the
        # in-between node)
        (118.517, 70.504), (118.21, 27.555)]


@pytest.mark.parametrize("mode", ["default", "fixed_point"])
@pytest.mark.parametrize("nx,kind,gd", CASES)
def test_fields_match_oracle(engine, nx, kind, gd, mode):
    engine.set_option("exact_ties", 1 if mode == "default" else 0)
    srcs = positions(nx, gd, FRAC)
    g, pv, veln, sols = oracle_case(nx, kind, gd, srcs)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
    assert (engine.nnx, engine.nnz) == (g.nnx, g.nnz)
    assert (bits(engine.velocity(0)) != bits(veln)).sum() == 0, "diced velocity grid must be bit-identical"
    n = len(srcs)
    # two receivers per source: far away and close to the source
    rcx = np.array([[srcs[(i + 3) % n][0], np.float32(s[0] + np.float32(0.3) * g.dnx)] for i, s in enumerate(srcs)], np.float32)
    rcz = np.array([[srcs[(i + 3) % n][1], np.float32(s[1] + np.float32(0.2) * g.dnz)] for i, s in enumerate(srcs)], np.float32)
    N = g.nnx
    gox, goz = g.gox, g.goz
    rcx = np.clip(rcx, gox, np.float32(gox + np.float32(N - 1.01) * g.dnx)).astype(np.float32)
    rcz = np.clip(rcz, goz, np.float32(goz + np.float32(N - 1.01) * g.dnz)).astype(np.float32)
    t = engine.traveltimes(np.zeros(n, np.int32), [s[0] for s in srcs], [s[1] for s in srcs], np.full(n, 2, np.int32),
                           rcx.reshape(-1), rcz.reshape(-1))
    worst = 0.0
    nbad_nodes = 0
    for u, (src, o) in enumerate(zip(srcs, sols)):
        T = engine.field(u)
        dT = np.abs(T - o["T"])
        d = float(dT.max())
        over = int((dT > TOL).sum())
        worst = max(worst, d)
        nbad_nodes += int((bits(T) != bits(o["T"])).sum())
        named = TIE_CASES.get((nx, kind, gd, u)) if mode == "fixed_point" else None
        if d > 0:
            parity_log.add(f"fixture nx={nx} {kind} gd={gd} source {u} {FRAC[u]} [{mode}]: field max |dT| {d:.9g} s, nodes beyond 1e-4 s {over}"
                           + (" [named tie case of the fixed point: reported, bounded]" if named is not None else ""))
        if named is not None:
            assert d <= named[0] and over <= named[1], (nx, kind, gd, u, d, over)
        else:
            assert d <= TOL and over == 0, (nx, kind, gd, u, d, over, mode)
        Tr, Sr = engine.refined(u)
        cls_o = np.sign(o["Sr"]).clip(-1, 1)
        # status classes may differ only at exact time ties (symmetric media); values where both alive agree
        both = (cls_o == 0) & (Sr == 0)
        assert np.abs(Tr[both] - o["Tr"][both]).max() <= TOL
        assert (cls_o != Sr).sum() <= 4, "refined status classes differ beyond tie noise"
        for k in range(2):
            ref = L.o_srtimes(g, veln, o["T"], src[0], src[1], rcx[u, k], rcz[u, k])
            assert abs(float(t[2 * u + k]) - float(ref)) <= TOL, (u, k, t[2 * u + k], ref)
    total = len(srcs) * g.nnx * g.nnz
    flagged = int(engine.stats()["tie_units"])
    parity_log.add(f"fixture nx={nx} {kind} gd={gd} [{mode}]: {len(srcs)} sources, worst field |dT| {worst:.3g} s, nodes not bit-identical {nbad_nodes} of {total}, units the census flagged {flagged}")
    assert nbad_nodes <= 0.001 * total


def test_sorted_variant_identical(engine):
    """the list variant of the solve kernel (option fim_sorted = 0; it serves the refined boxes only, the coarse solve runs on
    the compact field in the ordered variant) reaches the same fixed point as the tile-mask / ordered-sweep one, bit for bit"""
    nx = 35
    pv = np.stack([synth.medium(nx, "checker4"), synth.medium(nx, "smooth")])
    u = synth.units(nx, 12, 2, 6)
    out = []
    engine.set_option("exact_ties", 0)          # (the two variants of the fixed-point kernel are the subject)
    for v in (0, 1):
        engine.set_option("fim_sorted", v)
        engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        out.append((engine.traveltimes(**u), engine.field(23)))
    engine.set_option("fim_sorted", 1)          # back to the default
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
