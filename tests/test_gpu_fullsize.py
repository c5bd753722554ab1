"""Parity at BASELINE.json's full grid sizes.

* one unit per size against the oracle's Fast Marching field (the oracle needs ~0.3 s at 1025^2 and
  ~6 s at 4097^2, so one or two sources);
* a size-independent exact property for a whole batch at the headline size: halving every velocity
  doubles every travel time EXACTLY in fp32 (every operation of the path scales by a power of two:
  slowness, B-spline dicing, the stencil's quadratic, the source start-up, the receiver interpolation),
  so t(pv / 2) == 2 t(pv) bit for bit, whatever the schedule;
* reciprocity to discretisation accuracy, and exact repeatability of a batch.
"""
import numpy as np
import pytest

import os
from concurrent.futures import ThreadPoolExecutor

import _libs as L
import parity_log
import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# (nx, medium, period, band of the fixed point alone).  The DEFAULT mode (exact_ties = 1: fixed point, tie census, flagged units by the reference's
# march) must hold the north_star bar, 1e-4 s, over the whole field and at every receiver in every case.  The checkerboard of configs[4]
# (homogeneous blocks aligned with the grid) produces exact time ties between neighbouring narrow-band nodes; the reference's answer there depends
# on which of the two its heap pops first (CalSurfG.f90:417-424, :768-921) and its scheme carries the one-node difference downstream (DESIGN.md
# "Ties").  For those cases the fixed point alone (exact_ties = 0) is run beside the default and REPORTED, bounded by a band (largest |dT|, share of
# nodes beyond 1e-4 s) -- measured: 1025^2 4.43e-4 s on 14 nodes; 4097^2 (T up to 151 s, one ulp = 1.5e-5 s) 7.3e-4 s on 1.84 % of the nodes.
FULL = [(131, "smooth", 3, None), (131, "rough", 0, None), (131, "homog", 0, None),
        (131, "checker", 0, (8e-4, 1e-4)), (259, "checker", 1, None), (515, "checker", 2, (1.2e-3, 0.03))]


@pytest.mark.parametrize("nx,kind,period,band", FULL)
def test_one_unit_against_oracle(engine, nx, kind, period, band):
    """configs[2] (1025^2 smooth) and configs[4] (4097^2 checkerboard +-8 %, 16-vertex squares) media"""
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, kind, period)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    sx = np.float32(g.gox + np.float32(0.37 * (N - 1) + 0.3) * g.dnx)
    sz = np.float32(g.goz + np.float32(0.58 * (N - 1) + 0.6) * g.dnz)
    o = L.o_solve(g, pv, veln, sx, sz)
    rng = synth.LCG(nx)
    u = rng.uniform(64)
    rx = (g.gox + (0.5 + u[0::2] * (N - 2)).astype(np.float32) * g.dnx).astype(np.float32)
    rz = (g.goz + (0.5 + u[1::2] * (N - 2)).astype(np.float32) * g.dnz).astype(np.float32)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    assert (bits(engine.velocity(0)) != bits(veln)).sum() == 0
    ref = np.array([L.o_srtimes(g, veln, o["T"], sx, sz, rx[k], rz[k]) for k in range(32)], np.float32)
    t = engine.traveltimes([0], [sx], [sz], [32], rx, rz)          # the default mode
    flags, infl = engine.unit_ties()
    T = engine.field(0)
    d = np.abs(T - o["T"])
    line = (f"full N={N} {kind} [default mode{', unit marched' if flags[0] & 2 else ''}]: receivers max |dt| {np.abs(t - ref).max():.3g} s (32) | field max {d.max():.9g} s, "
            f"beyond 1e-4 s {int((d > TOL).sum())} nodes, not bit-identical {100.0 * (bits(T) != bits(o['T'])).mean():.3f} %, largest tie influence {infl[0]:.3g} s")
    assert np.abs(t - ref).max() <= TOL
    assert d.max() <= TOL and int((d > TOL).sum()) == 0
    if band is not None:          # a named tie case: the fixed point alone, reported and bounded
        assert flags[0] & 2, "the tie census must flag this unit"
        engine.set_option("exact_ties", 0)
        t0 = engine.traveltimes([0], [sx], [sz], [32], rx, rz)
        T0 = engine.field(0)
        d0 = np.abs(T0 - o["T"])
        st0 = engine.stats()
        line += (f" | fixed point alone [named tie case]: receivers max |dt| {np.abs(t0 - ref).max():.3g} s, field max {d0.max():.9g} s, beyond 1e-4 s {int((d0 > TOL).sum())} nodes = "
                 f"{100.0 * (d0 > TOL).mean():.4f} %, census: {int(st0['tie_units'])} unit flagged and left to the fixed point")
        assert d0.max() <= band[0] and (d0 > TOL).mean() <= band[1]
        assert st0["tie_units"] == 1 and st0["tie_units_left"] == 1 and st0["exact_units"] == 0
    parity_log.add(line)


def test_config1_homogeneous_256(engine):
    """BASELINE.json configs[1]: 257^2 grid, pv = 3.0 km/s, 64 sources in the inner 90 %, receivers = all
    later sources (2016 pairs): FIM against FMM travel times"""
    nx, nsrc = 35, 64
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, "homog")
    veln = L.o_gridder(g, pv)
    sx, sz = synth.sources(nx, nsrc)
    nrec = np.array([nsrc - 1 - i for i in range(nsrc)], np.int32)
    rx = np.concatenate([sx[i + 1:] for i in range(nsrc)]).astype(np.float32)
    rz = np.concatenate([sz[i + 1:] for i in range(nsrc)]).astype(np.float32)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t = engine.traveltimes(np.zeros(nsrc, np.int32), sx, sz, nrec, rx, rz)
    assert t.size == 2016
    ref = np.zeros_like(t)
    k = 0
    for i in range(nsrc):
        o = L.o_solve(g, pv, veln, sx[i], sz[i])
        for j in range(i + 1, nsrc):
            ref[k] = L.o_srtimes(g, veln, o["T"], sx[i], sz[i], sx[j], sz[j])
            k += 1
    assert np.abs(t - ref).max() <= TOL


def test_scaling_is_exact_at_headline_size(engine):
    """512 receiver times of 16 units at 1025^2: t(pv / 2) == 2 * t(pv), bit for bit"""
    nx = 131
    u = synth.units(nx, 8, 2, 32)
    pv = np.stack([synth.medium(nx, "smooth", p) for p in range(2)])
    engine.set_option("exact_ties", 0)          # (a property of the fixed point's arithmetic: the tie threshold is in seconds and does not scale with the medium)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t1 = engine.traveltimes(**u)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 0.5 * pv)
    t2 = engine.traveltimes(**u)
    assert t1.size == 512 and np.isfinite(t1).all() and (t1 >= 0).all() and (t1 > 0).sum() >= 448     # (a receiver may coincide with its source)
    assert (bits(t2) != bits(np.float32(2.0) * t1)).sum() == 0
    # repeatability: the same batch again, bit for bit
    t3 = engine.traveltimes(**u)
    assert (bits(t3) != bits(t2)).sum() == 0
    # reciprocity (a property of the continuum problem): source and receiver swapped agree to the
    # discretisation error of a 139 m grid, far below a per cent
    sx, sz = u["scx"][:8], u["scz"][:8]
    nrec = np.full(8, 7, np.int32)
    rx = np.concatenate([np.delete(sx, i) for i in range(8)]).astype(np.float32)
    rz = np.concatenate([np.delete(sz, i) for i in range(8)]).astype(np.float32)
    t = engine.traveltimes(np.zeros(8, np.int32), sx, sz, nrec, rx, rz).reshape(8, 7)
    full = np.zeros((8, 8))
    for i in range(8):
        full[i, np.arange(8) != i] = t[i]
    off = ~np.eye(8, dtype=bool)
    assert np.abs(full - full.T)[off].max() <= 0.01 * full[off].min() + 0.02


@pytest.mark.parametrize("kind", ["smooth", "rough"])
def test_receivers_at_scale(engine, kind):
    """256 units x 32 receivers at the headline size (1025^2; smooth = configs[2]'s medium, rough = +-10 % random vertices)
    against the oracle's Fast Marching (pinned to the reference at this size: tests/test_oracle_vs_ref.py), the oracle
    solves spread over the host cores: every one of the 8192 receiver times within 1e-4 s"""
    nx, nsrc, nper, nrec = 131, 128, 2, 32
    u = synth.units(nx, nsrc, nper, nrec)
    pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t = engine.traveltimes(**u).reshape(nsrc * nper, nrec)
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    veln = [L.o_gridder(g, pv[p]) for p in range(nper)]

    def one(k):
        p = int(u["map_index"][k])
        o = L.o_solve(g, pv[p], veln[p], u["scx"][k], u["scz"][k])
        return np.array([L.o_srtimes(g, veln[p], o["T"], u["scx"][k], u["scz"][k], u["rcx"][k * nrec + r], u["rcz"][k * nrec + r]) for r in range(nrec)], np.float32)

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:      # ctypes releases the GIL; the oracle keeps no global state
        ref = np.stack(list(ex.map(one, range(nsrc * nper))))
    d = np.abs(t.astype(np.float64) - ref.astype(np.float64))
    parity_log.add(f"receivers at scale N=1025 {kind}: {d.size} receiver times of {nsrc * nper} units, max |dt| {d.max():.3g} s, beyond 1e-4 s {int((d > TOL).sum())}, "
                   f"not bit-identical {int((bits(t) != bits(ref)).sum())}")
    assert d.max() <= TOL


def test_wild_medium_keeps_the_exception_table_small(engine):
    """+-45 % random vertices at 1025^2: several hundred non-causal nodes (tau != T) along colliding fronts, i.e. the worst case for
    the exception table of the compact coarse field (capacity 16 384 here).  The solve must go through, the receivers must hold the
    bar, and the field may differ from Fast Marching only by tie noise."""
    nx = 131
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, "wild")
    veln = L.o_gridder(g, pv)
    N = g.nnx
    sx = np.float32(g.gox + np.float32(0.41 * (N - 1) + 0.3) * g.dnx)
    sz = np.float32(g.goz + np.float32(0.52 * (N - 1) + 0.6) * g.dnz)
    o = L.o_solve(g, pv, veln, sx, sz)
    u = synth.LCG(77).uniform(64)
    rx = (g.gox + (0.5 + u[0::2] * (N - 2)).astype(np.float32) * g.dnx).astype(np.float32)
    rz = (g.goz + (0.5 + u[1::2] * (N - 2)).astype(np.float32) * g.dnz).astype(np.float32)
    engine.set_option("exact_ties", 0)          # (the fixed point's exception table is the subject)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t = engine.traveltimes([0], [sx], [sz], [32], rx, rz)
    ref = np.array([L.o_srtimes(g, veln, o["T"], sx, sz, rx[k], rz[k]) for k in range(32)], np.float32)
    T = engine.field(0)
    tau = engine.debug_field(0, 1)
    d = np.abs(T - o["T"])
    nexc = int((np.abs(tau) != T).sum())
    parity_log.add(f"wild N={N}: receivers max |dt| {np.abs(t - ref).max():.3g} s | field max {d.max():.3g} s, beyond 1e-4 s {int((d > TOL).sum())} nodes | nodes with tau != T: {nexc}")
    assert np.abs(t - ref).max() <= TOL
    assert nexc >= 100                      # the case does exercise the table
    assert d.max() <= 2e-3 and (d > TOL).mean() <= 1e-4
    # a table that is far too small (256 entries) overflows, grows by itself (x4 per attempt) and gives the same field
    engine.set_option("exc_log2cap", 8)
    try:
        engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        t2 = engine.traveltimes([0], [sx], [sz], [32], rx, rz)
        T2 = engine.field(0)
        grown = engine.stats()["rescans"]
    finally:
        engine.set_option("exc_log2cap", 0)
    assert grown >= 1
    assert np.array_equal(bits(t2), bits(t)) and np.array_equal(bits(T2), bits(T))


def test_recycled_field_slots_give_the_same_times(engine):
    """option field_pool: a launch with more units than coarse field slots hands the slots from workgroup to workgroup (each resets
    its slot, solves, writes its unit's receiver times itself, then releases the slot).  Same bits as one slot per unit; and the
    fields of a recycled solve are reported as gone instead of being read from a slot another unit has reused."""
    from dsurftomo_amd.engine import EngineError
    nx, nsrc, nper, nrec = 35, 300, 2, 6
    u = synth.units(nx, nsrc, nper, nrec)
    pv = np.stack([synth.medium(nx, k, p) for p, k in enumerate(("checker4", "rough"))])
    out = {}
    engine.set_option("bundle", 0)              # (the unit-by-unit kernel's slot pool is the subject: since round 4 a 257^2 grid would bundle the two periods of a source)
    engine.set_option("exact_ties", 0)
    try:
        for pool in (-1, 16, 97, 0):
            engine.set_option("field_pool", pool)
            engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
            out[pool] = engine.traveltimes(**u)
            st = engine.stats()
            assert st["field_slots"] == (600 if pool <= 0 else pool)
            if pool > 0:
                with pytest.raises(EngineError):
                    engine.field(0)
            else:
                engine.field(599)
    finally:
        engine.set_option("field_pool", 0)
        engine.set_option("bundle", 1)
    assert np.isfinite(out[-1]).all() and (out[-1] > 0).sum() > 0.95 * out[-1].size
    for pool in (16, 97, 0):
        assert np.array_equal(bits(out[pool]), bits(out[-1])), pool
    parity_log.add(f"field slots: 600 units at N=257 through 16 / 97 / 600 slots: receiver times bit-identical")


# ---------------------------------------------------------------------------------------------------------------------------------
# Tie-prone media at full size, all three modes (tests of the bar, not pins of a miss).  The DEFAULT mode (exact_ties = 1)
# must hold 1e-4 s; exact_ties = 2 (every unit by the reference's march) must be bit-identical; exact_ties = 0 (the fixed point alone) differs
# from the reference where its heap decided an exact tie (DESIGN.md "Ties") and is reported, bounded by a band around what was measured:
#   configs[4]'s medium at 4097^2, 128 units x 32 receivers: 139 of 4096 times beyond 1e-4 s, worst 6.9e-4 s (times up to 217.6 s);
#   1025^2 rough, 64 random sources: 33-37 of 64 fields with a node beyond 1e-4 s, worst node 1.27e-3 s, ~640 of 67.2 M nodes.
BANDS = {"config4_receivers": (1.2e-3, 0.08), "rough1025_fields": (3e-3, 1e-4)}


def _oracle_receivers(nx, pv, u, nrec, units):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    veln = {p: L.o_gridder(g, pv[p]) for p in sorted(set(int(u["map_index"][k]) for k in units))}

    def one(k):
        p = int(u["map_index"][k])
        o = L.o_solve(g, pv[p], veln[p], u["scx"][k], u["scz"][k])
        return np.array([L.o_srtimes(g, veln[p], o["T"], u["scx"][k], u["scz"][k], u["rcx"][k * nrec + r], u["rcz"][k * nrec + r]) for r in range(nrec)], np.float32)

    with ThreadPoolExecutor(max_workers=min(24, os.cpu_count() or 1)) as ex:
        return np.stack(list(ex.map(one, units)))


def test_receivers_at_scale_config4(engine):
    """configs[4]'s grid and medium (4097^2, checkerboard +-8 %, 16-vertex squares): 128 units x 32 receivers against the oracle's
    Fast Marching.  Default mode: within 1e-4 s.  exact_ties = 2: bit-identical.  exact_ties = 0: reported, inside its band."""
    nx, nsrc, nper, nrec = 515, 64, 2, 32
    u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + 11)
    pv = np.stack([synth.medium(nx, "checker", p) for p in range(nper)])
    n = nsrc * nper
    ref = _oracle_receivers(nx, pv, u, nrec, range(n))
    engine.set_option("max_chunk", 256)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t1 = engine.traveltimes(**u).reshape(n, nrec)          # the default mode
    st1 = engine.stats()
    flags1, _ = engine.unit_ties()
    engine.set_option("exact_ties", 2)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    tx = engine.traveltimes(**u).reshape(n, nrec)
    st = engine.stats()
    engine.set_option("exact_ties", 0)
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    t = engine.traveltimes(**u).reshape(n, nrec)
    st0 = engine.stats()
    d = np.abs(t.astype(np.float64) - ref.astype(np.float64))
    d1 = np.abs(t1.astype(np.float64) - ref.astype(np.float64))
    marched = (flags1 & 2) != 0
    beyond, worst = int((d > TOL).sum()), "%.9g" % d.max()
    parity_log.add(f"configs[4] medium N=4097, {n} units x {nrec} receivers: default mode (exact_ties=1): {int(marched.sum())} of {n} units flagged and marched, receiver times beyond 1e-4 s "
                   f"{int((d1 > TOL).sum())}, max |dt| {d1.max():.3g} s ({st1['exact_pops'] / max(st1['ms_exact'], 1e-9) / 1e3:.0f} M accepts/s) | exact_ties=2: not bit-identical "
                   f"{int((bits(tx) != bits(ref)).sum())}, {st['exact_pops'] / max(st['ms_exact'], 1e-9) / 1e3:.0f} M accepts/s | fixed point alone (exact_ties=0) [reported]: max |dt| {worst} s, "
                   f"beyond 1e-4 s {beyond} of {d.size}, not bit-identical {int((bits(t) != bits(ref)).sum())} (times up to {ref.max():.1f} s), census: {int(st0['tie_units'])} units flagged, all left")
    assert (d1 > TOL).sum() == 0                                         # the default mode holds north_star's tolerance on configs[4]'s medium
    assert (bits(t1[marched]) != bits(ref[marched])).sum() == 0          # flagged units: the reference's bits
    assert (bits(tx) != bits(ref)).sum() == 0
    assert d.max() <= BANDS["config4_receivers"][0] and beyond <= BANDS["config4_receivers"][1] * d.size
    assert st0["tie_units_left"] == st0["tie_units"] and st0["exact_units"] == 0


def test_fields_at_headline_size_rough(engine):
    """1025^2, rough +-10 % medium, 64 random sources (a quarter of them on node lines, as tests/tools/fuzz_parity.py draws them):
    whole fields against the oracle.  Default mode: every field within 1e-4 s, flagged units bit-identical; exact_ties = 2: every field
    bit-identical; exact_ties = 0: reported, inside its band."""
    nx, nsrc = 131, 64
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    N = g.nnx
    pv = synth.medium(nx, "rough", 0)
    veln = L.o_gridder(g, pv)
    r = synth.LCG(synth.SEED + 23).uniform(2 * nsrc)
    fx = 0.5 + r[0::2] * (N - 2.0)
    fz = 0.5 + r[1::2] * (N - 2.0)
    fx[::4] = np.round(fx[::4]); fz[1::8] = np.round(fz[1::8])
    sx = (g.gox + fx.astype(np.float32) * g.dnx).astype(np.float32)
    sz = (g.goz + fz.astype(np.float32) * g.dnz).astype(np.float32)
    with ThreadPoolExecutor(max_workers=min(24, os.cpu_count() or 1)) as ex:
        sols = list(ex.map(lambda k: L.o_solve(g, pv, veln, sx[k], sz[k])["T"], range(nsrc)))
    engine.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    rx = np.repeat(sx[::-1], 1).astype(np.float32); rz = np.repeat(sz[::-1], 1).astype(np.float32)
    args = (np.zeros(nsrc, np.int32), sx, sz, np.ones(nsrc, np.int32), rx, rz)
    ref1 = np.array([L.o_srtimes(g, veln, sols[k], sx[k], sz[k], rx[k], rz[k]) for k in range(nsrc)], np.float32)
    t1 = engine.traveltimes(*args)                       # the default mode
    flags1, _ = engine.unit_ties()
    marched = (flags1 & 2) != 0
    dm1 = np.array([np.abs(engine.field(k) - sols[k]).max() for k in range(nsrc)])
    bad1 = sum(int((bits(engine.field(k)) != bits(sols[k])).sum()) for k in range(nsrc) if marched[k])
    engine.set_option("exact_ties", 2)
    engine.traveltimes(*args)
    exact_bad = sum(int((bits(engine.field(k)) != bits(sols[k])).sum()) for k in range(nsrc))
    engine.set_option("exact_ties", 0)
    engine.traveltimes(*args)
    dm = np.array([np.abs(engine.field(k) - sols[k]).max() for k in range(nsrc)])
    nbeyond = np.array([int((np.abs(engine.field(k) - sols[k]) > TOL).sum()) for k in range(nsrc)])
    fields_bad, worst = int((dm > TOL).sum()), "%.9g" % dm.max()
    parity_log.add(f"N=1025 rough, {nsrc} random sources: default mode (exact_ties=1): {int(marched.sum())} of {nsrc} units flagged and marched (nodes not bit-identical in them {bad1}), "
                   f"worst node of all fields {dm1.max():.3g} s, of the fields left to the fixed point {dm1[~marched].max() if (~marched).any() else 0.0:.3g} s, receiver times beyond 1e-4 s "
                   f"{int((np.abs(t1 - ref1) > TOL).sum())} | exact_ties=2: nodes not bit-identical {exact_bad} | fixed point alone (exact_ties=0) [reported]: {fields_bad} fields with a node beyond "
                   f"1e-4 s (worst node {worst} s, {int(nbeyond.sum())} nodes of {nsrc * N * N} beyond)")
    assert exact_bad == 0
    assert bad1 == 0 and (np.abs(t1 - ref1) > TOL).sum() == 0
    assert (dm1 <= TOL).all()                            # the default mode: every node of every field
    worst_cap, share_cap = BANDS["rough1025_fields"]
    assert dm.max() <= worst_cap and nbeyond.sum() <= share_cap * nsrc * N * N
