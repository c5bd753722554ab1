"""Whole-boundary parity on the GPU: the drop-in entries dsa_calsurfg / dsa_synthetic (C ABI), the
Fortran shim, and the dispersion stage, against the oracle and the committed golden vectors.

Tolerances.  Travel times: 1e-4 s (north_star).  The eikonal solve can differ from Fast Marching
only where two arrivals tie to the last bit, so rays, rows and dispersion -- which run
the reference's arithmetic operation for operation -- are bit-identical on every case of this file
(measured: tests/tools/parity_table.py), and that is what is asserted: receiver times and every matrix
entry equal to the oracle's, bit for bit (no case of this file is a tie case).  Dispersion runs in fp64 with the
device's sin / cos / exp instead of libm's; a last-bit difference there can move an fp32-rounded
phase velocity by one ulp (2.4e-7), i.e. a depth kernel by 2.4e-7 / (0.01 v) ~ 1e-5.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import _libs as L
import parity_log
import synth
from dsurftomo_amd import io as taipei      # the reference's Taipei example (tests/golden/taipei/) through the package's format readers

pytestmark = pytest.mark.gpu

CASES = {
    "default": dict(),
    "deep": dict(deep=True, nz=7, nx=10, ny=13, seed=5),
    "groups": dict(kRc=0, kRg=2, kLc=0, kLg=1),
    "big": dict(nx=20, ny=18, nz=6, nsrc=8, nrcf=7, kRc=3, kRg=1, kLc=1, kLg=1),
}


@pytest.fixture(scope="module")
def lib():
    from dsurftomo_amd import engine
    return engine.load_library()


def dense(r, ndata, npar):
    G = np.zeros((ndata, npar), np.float32)
    G[r["iw"] - 1, r["col"] - 1] = r["rw"]
    return G


def check_rows(o, d, c, name=""):
    Go, Gd = dense(o, c["ndata"], c["nparpi"]), dense(d, c["ndata"], c["nparpi"])
    differ = Go.view(np.uint32) != Gd.view(np.uint32)
    tdiff = int((o["dsurf"].view(np.uint32) != d["dsurf"].view(np.uint32)).sum())
    parity_log.add("boundary %s: %d data, max |d dsurf| %.3g s (not bit-identical %d); nar %d vs %d, matrix entries differing %d (max %.3g)" %
                   (name, c["ndata"], np.abs(o["dsurf"] - d["dsurf"]).max(), tdiff, o["nar"], d["nar"], int(differ.sum()), np.abs(Go - Gd).max()))
    assert np.abs(o["dsurf"] - d["dsurf"]).max() <= 1e-4
    assert differ.sum() == 0, "%d of %d matrix entries differ" % (differ.sum(), o["nar"])
    assert o["nar"] == d["nar"]
    # the reference's order: rows ascending, columns ascending inside a row
    key = d["iw"].astype(np.int64) * (c["nparpi"] + 1) + d["col"]
    assert (np.diff(key) > 0).all()
    return int(differ.sum())


@pytest.mark.parametrize("name", sorted(CASES))
def test_dropin_calsurfg_and_synthetic(lib, name):
    c = synth.boundary_case(**CASES[name])
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    d = L.call_boundary(lib.dsa_calsurfg, c)
    assert lib.dsa_dropin_error() == b"" or d["nar"] > 0, lib.dsa_dropin_error()
    check_rows(o, d, c, name)
    so = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)
    sd = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
    assert np.abs(so - sd).max() <= 1e-4


def test_taipei_example(lib):
    """BASELINE.json configs[0]: first forward call of the Taipei inversion (18x18x9 model, 26
    Rayleigh phase periods, 2061 data) against the reference's own output (golden) and the oracle"""
    c = taipei.load()
    d = L.call_boundary(lib.dsa_calsurfg, c)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "b_taipei.npz"))
    assert np.abs(d["dsurf"] - z["dsurf"]).max() <= 1e-4
    assert d["nar"] == int(z["nar"])
    G = dense(d, c["ndata"], c["nparpi"])
    assert np.abs(G.sum(axis=1, dtype=np.float64) - z["row_sums"]).max() <= 5e-3
    assert np.abs(np.abs(G).sum(axis=0, dtype=np.float64) - z["col_abs_sums"]).max() <= 2e-2
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    check_rows(o, d, c, "taipei")


@pytest.mark.parametrize("iwave,igr", [(2, 0), (2, 1), (1, 0), (1, 1)])
def test_dispersion_stage(iwave, igr):
    from dsurftomo_amd.engine import Engine
    c = synth.boundary_case(nx=12, ny=10, nz=7)
    vel = np.ascontiguousarray(c["vels"].T)
    t = np.array([1.0, 2.0, 4.0, 7.0, 11.0, 16.0, 22.0])
    ref = L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), iwave, igr, t)
    e = Engine(0)
    try:
        e.dispersion_begin(vel, c["depz"], float(c["minthk"]), len(t), len(t))
        e.dispersion_run(iwave, igr, t, True, 0, 0)
        dev = e.dispersion_fetch(0, len(t), True, 0)
    finally:
        e.close()
    assert np.abs(dev[0] - ref[0]).max() <= 5e-7
    assert (dev[0] == ref[0]).mean() >= 0.99
    for a, b in zip(dev[1:], ref[1:]):
        assert np.abs(a - b).max() <= 5e-5
        assert (a == b).mean() >= 0.98


@pytest.mark.parametrize("max_chunk,ray_budget,ray_lanes", [(0, 0, 0), (2, 40000, 0), (0, 0, 1), (0, 0, 4)])
def test_engine_rows_from_host_kernels(max_chunk, ray_budget, ray_lanes):
    """engine level: maps and depth kernels handed over from the host (dsa_set_maps +
    dsa_set_depth_kernels + dsa_plan_units + dsa_solve_rows) instead of the device dispersion stage;
    second variant: two units per chunk and a ray budget of a few rays per launch, so that the unit
    chunks and the ray launches are stitched many times; then the tracer with one lane per ray and with
    four lanes per ray (the engine picks by the size of the launch): the same rows, bit for bit"""
    from dsurftomo_amd.engine import Engine
    c = synth.boundary_case(kRc=3, kRg=0, kLc=0, kLg=0)
    vel = np.ascontiguousarray(c["vels"].T)
    pv, svs, svp, srho = L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), 2, 0, c["tRc"])
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    maps, sx, sz, nrec, rx, rz, slot = [], [], [], [], [], [], []
    for k in range(c["kmax"]):
        for s in range(c["nsrcsurf1"][k]):
            maps.append(c["periods"][s, k] - 1); sx.append(c["scxf"][s, k]); sz.append(c["sczf"][s, k]); slot.append(k)
            nrec.append(c["nrc1"][s, k])
            rx += list(c["rcxf"][:nrec[-1], s, k]); rz += list(c["rczf"][:nrec[-1], s, k])
    e = Engine(0)
    try:
        e.set_option("max_chunk", max_chunk)
        e.set_option("ray_budget", ray_budget)
        e.set_option("ray_lanes", ray_lanes)
        e.set_maps(c["nx"], c["ny"], c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], pv)
        e.set_depth_kernels(vel, c["depz"], svs, svp, srho)
        e.plan(maps, sx, sz, nrec, rx, rz, sen_slot=slot)
        t, rw, iw, col = e.solve_rows(c["ndata"] * c["nparpi"])
        st = e.stats()
    finally:
        e.close()
    check_rows(o, dict(dsurf=t, rw=rw, iw=iw, col=col, nar=rw.size), c)
    assert st["rays"] == c["ndata"] and st["nar"] == rw.size


@pytest.mark.parametrize("ray_budget", [0, 40000])
def test_ray_paths_against_oracle(ray_budget, tmp_path):
    """the points of every traced ray (option ray_path_cap; what the reference's disabled raypath.out dump would hold,
    CalSurfG.f90:2276-2283) against the oracle's ray tracer, bit for bit; second variant: a few rays per launch"""
    from dsurftomo_amd.engine import Engine
    from dsurftomo_amd import io
    c = synth.boundary_case(kRc=2, kRg=0, kLc=0, kLg=0)
    vel = np.ascontiguousarray(c["vels"].T)
    pv, svs, svp, srho = L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), 2, 0, c["tRc"])
    maps, sx, sz, nrec, rx, rz, slot = [], [], [], [], [], [], []
    for k in range(c["kmax"]):
        for s in range(c["nsrcsurf1"][k]):
            maps.append(c["periods"][s, k] - 1); sx.append(c["scxf"][s, k]); sz.append(c["sczf"][s, k]); slot.append(k)
            nrec.append(c["nrc1"][s, k])
            rx += list(c["rcxf"][:nrec[-1], s, k]); rz += list(c["rczf"][:nrec[-1], s, k])
    cap = 4096
    e = Engine(0)
    try:
        e.set_option("ray_budget", ray_budget)
        e.set_option("ray_path_cap", cap)
        e.set_maps(c["nx"], c["ny"], c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], pv)
        e.set_depth_kernels(vel, c["depz"], svs, svp, srho)
        e.plan(maps, sx, sz, nrec, rx, rz, sen_slot=slot)
        e.solve_rows(c["ndata"] * c["nparpi"])
        paths = e.ray_paths(cap)
    finally:
        e.close()
    assert [d for d, _ in paths] == list(range(1, c["ndata"] + 1))
    g = L.grid(c["nx"], c["ny"], c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], 8)
    r = 0
    for u in range(len(maps)):
        veln = L.o_gridder(g, pv[maps[u]])
        sol = L.o_solve(g, pv[maps[u]], veln, sx[u], sz[u])
        for q in range(nrec[u]):
            want = L.o_ray_path(g, sol, veln, sx[u], sz[u], rx[r], rz[r])
            got = paths[r][1]
            assert got.shape == want.shape and (got.view(np.uint32) == want.view(np.uint32)).all(), (u, q)
            assert len(got) >= 2
            r += 1
    # receiver first, source last, in degrees
    lat0 = 90.0 - np.degrees(rx[0]); lon0 = np.degrees(rz[0])
    assert abs(paths[0][1][0, 0] - lat0) < 1e-4 and abs(paths[0][1][0, 1] - lon0) < 1e-4
    assert abs(paths[0][1][-1, 0] - (90.0 - np.degrees(sx[0]))) < 1e-4
    # the reference's file format: '# nrp' then one 'lat lon' line per point (its scripts/plotpath.py reads that)
    out = str(tmp_path / "raypath.out")
    io.write_raypaths(out, paths)
    lines = open(out).read().split("\n")
    assert lines[0].split() == ["#", str(len(paths[0][1]))] and len(lines[1].split()) == 2
    assert sum(1 for l in lines if l.lstrip().startswith("#")) == len(paths)


def write_driver_input(c, fin, maxnar):
    with open(fin, "wb") as f:
        np.array([c["nx"], c["ny"], c["nz"], c["kRc"], c["kRg"], c["kLc"], c["kLg"], c["kmax"], c["nsrcsurf"], c["nrcf"], c["ndata"], maxnar], np.int32).tofile(f)
        np.array([c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], c["minthk"]], np.float32).tofile(f)
        for a in (c["vels"], c["depz"], c["tRc"], c["tRg"], c["tLc"], c["tLg"], c["wavetype"], c["igrt"], c["periods"], c["nrc1"], c["nsrcsurf1"],
                  c["scxf"], c["sczf"], c["rcxf"], c["rczf"]):
            f.write(np.asarray(a).tobytes(order="F"))


def diagnostics_case():
    """a call that makes the reference speak: one source / receiver pair hugging the northern edge under a medium that gets faster
    towards it (the ray leaves the grid and is clamped: rbint, CalSurfG.f90:2082-2101), and one grid column without any velocity
    contrast (no Love-wave root: surfdisp96.f:308-339)"""
    c = synth.boundary_case(nx=12, ny=11, nz=5, kRc=2, kRg=1, kLc=1, kLg=0, nsrc=3, nrcf=3, ragged=False)
    f = np.float32
    lat = float(c["goxd"]) - 0.004 * float(c["dvxd"])
    lon0, lon1 = float(c["gozd"]) + 0.4 * float(c["dvzd"]), float(c["gozd"]) + (c["ny"] - 3.4) * float(c["dvzd"])
    s, k = 1, 1                          # second source of the second period slot: iteration 4 of 12 (0-based)
    c["scxf"][s, k] = f((90.0 - lat) * np.pi / 180.0); c["sczf"][s, k] = f(lon0 * np.pi / 180.0)
    c["rcxf"][0, s, k] = f((90.0 - lat) * np.pi / 180.0); c["rczf"][0, s, k] = f(lon1 * np.pi / 180.0)
    v = np.array(c["vels"])
    v *= (1.0 + 0.25 * np.exp(-np.arange(c["nx"])[:, None, None] / 2.0)).astype(np.float32)
    v[6, 5, :] = v[6, 5, 0]
    c["vels"] = np.asfortranarray(v.astype(np.float32))
    return c


# what the reference itself wrote for diagnostics_case() (its calsurfg_, built by oracle/Makefile, run in the build container by the
# reference leg of the test below, which repeats the comparison wherever oracle/_ref is present): boundary notes on unit 6, blocks on unit 66
REF_NOTES, REF_BLOCKS = 8, 3


def test_fortran_shim_prints_the_reference_diagnostics(tmp_path):
    """non-fatal diagnostics of the boundary: the shim writes the reference's boundary note to unit 6 as many times as
    the reference does, and the 'no zero found' text to unit 66 with the number of dispersion curves that ended that way"""
    exe = os.path.join(L.ROOT, "tests", "build", "shim_driver")
    if not os.path.exists(exe):
        pytest.skip("tests/build/shim_driver was not built (flang missing)")
    c = diagnostics_case()
    maxnar = c["ndata"] * c["nparpi"]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_driver_input(c, fin, maxnar)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.exists(fout), r.stdout + r.stderr
    notes = r.stdout.count("Note that at least one two-point ray path")
    log = (tmp_path / "fort.66").read_text() if (tmp_path / "fort.66").exists() else ""
    ncurves = [int(line.split(":")[1]) for line in log.splitlines() if "curves of this call that ended this way" in line]
    assert "improper initial value in disper - no zero found" in log and "(1=L, 2=R)" in log and "due to looking for Love waves in a halfspace" in log
    assert notes == REF_NOTES and ncurves == [REF_BLOCKS], (notes, ncurves, log[-600:])
    # the numbers still come out as the oracle's (the clamped ray included)
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    with open(fout, "rb") as f:
        nar = int(np.fromfile(f, np.int32, 1)[0])
        dsurf = np.fromfile(f, np.float32, c["ndata"]); np.fromfile(f, np.float32, c["ndata"])
        rw = np.fromfile(f, np.float32, nar); iw = np.fromfile(f, np.int32, nar); col = np.fromfile(f, np.int32, nar)
    check_rows(o, dict(dsurf=dsurf, rw=rw, iw=iw, col=col, nar=nar), c)
    if L.ref() is not None:              # the reference's own words, same call, in a process of its own (Fortran units 6 and 66)
        refdir = tmp_path / "ref"
        refdir.mkdir()
        code = ("import sys; sys.path[:0] = [%r, %r]; import _libs as L, test_gpu_boundary as t; "
                "L.call_boundary(L.ref().calsurfg_, t.diagnostics_case())" % (L.ROOT, os.path.join(L.ROOT, "tests")))
        rr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(refdir))
        assert rr.returncode == 0, rr.stderr[-800:]
        assert rr.stdout.count("Note that at least one two-point ray path") == notes
        assert (refdir / "fort.66").read_text().count("improper initial value in disper - no zero found") == ncurves[0]
        # round 4 (missing 5): with DSA_DISP_FAILURE_LOG the shim writes the reference's block once per failing surfdisp96
        # call, layer table and all -- the reference's unit-66 file of the same call on one thread, character for character
        ref1 = tmp_path / "ref1"
        ref1.mkdir()
        rr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ref1), env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert rr.returncode == 0, rr.stderr[-800:]
        logdir = tmp_path / "log"
        logdir.mkdir()
        r2 = subprocess.run([exe, fin, str(logdir / "out.bin")], capture_output=True, text=True, timeout=300, cwd=str(logdir), env=dict(os.environ, DSA_DISP_FAILURE_LOG="16"))
        assert r2.returncode == 0, r2.stdout + r2.stderr
        ours, theirs = (logdir / "fort.66").read_text(), (ref1 / "fort.66").read_text()
        assert "d,a,b,rho (d(mmax)=control ignore)" in ours and ours.count("improper initial value") == REF_BLOCKS
        assert ours == theirs, "unit 66 differs from the reference's:\n" + "\n".join(a + "   |   " + b for a, b in zip(ours.splitlines(), theirs.splitlines()) if a != b)[:2000]
        parity_log.add(f"unit 66 with DSA_DISP_FAILURE_LOG: {ours.count('improper initial value')} blocks of {len(ours.splitlines()) // max(ours.count('improper initial value'), 1)} lines (layer tables included), identical to the reference's file of the same call: {ours == theirs}")


def test_fortran_shim(tmp_path):
    """calsurfg_ / synthetic_ through the flang-built shim and a Fortran caller"""
    exe = os.path.join(L.ROOT, "tests", "build", "shim_driver")
    if not os.path.exists(exe):
        pytest.skip("tests/build/shim_driver was not built (flang missing)")
    c = synth.boundary_case()
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    so = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)
    maxnar = c["ndata"] * c["nparpi"]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_driver_input(c, fin, maxnar)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.exists(fout), r.stdout + r.stderr
    with open(fout, "rb") as f:
        nar = int(np.fromfile(f, np.int32, 1)[0])
        dsurf = np.fromfile(f, np.float32, c["ndata"]); obst = np.fromfile(f, np.float32, c["ndata"])
        rw = np.fromfile(f, np.float32, nar); iw = np.fromfile(f, np.int32, nar); col = np.fromfile(f, np.int32, nar)
        xv = np.fromfile(f, np.float32, c["nparpi"]); yv = np.fromfile(f, np.float32, c["ndata"])
        istop, itn = np.fromfile(f, np.int32, 2)
        dv = np.fromfile(f, np.float32, c["nparpi"]); est = np.fromfile(f, np.float32, 5)
    # aprod_ through aprod_shim.f90: y += A x then x += A^T y, against the oracle's aprod on the same matrix
    x0 = (((np.arange(1, c["nparpi"] + 1) * 7) % 13 - 6) * 0.125).astype(np.float32)
    y0 = (((np.arange(1, c["ndata"] + 1) * 5) % 11 - 5) * 0.25).astype(np.float32)
    iwf = np.concatenate([[nar], iw, col]).astype(np.int32)
    ib = lambda v: C.byref(C.c_int(int(v)))
    for mode in (1, 2):
        L.oracle().dso_aprod(ib(mode), ib(c["ndata"]), ib(c["nparpi"]), L.ptr(x0), L.ptr(y0), ib(iwf.size), ib(nar), L.ptr(iwf), L.ptr(rw))
    assert (x0.view(np.uint32) != xv.view(np.uint32)).sum() == 0 and (y0.view(np.uint32) != yv.view(np.uint32)).sum() == 0
    # LSMR through lsmr_shim.f90 (module lsmrModule) against the oracle's LSMR on the same matrix
    import inversion as inv
    bv = (((np.arange(1, c["ndata"] + 1) * 3) % 17 - 8) * np.float32(0.01)).astype(np.float32)
    want = inv.call_lsmr(L.oracle().dso_lsmr, dict(m=c["ndata"], n=c["nparpi"], iw=iwf, rw=rw, b=bv), 1.0)
    got = dict(x=dv, istop=int(istop), itn=int(itn), normA=est[0], condA=est[1], normr=est[2], normAr=est[3], normx=est[4])
    assert want["itn"] > 3 and inv.same(got, want) == []
    check_rows(o, dict(dsurf=dsurf, rw=rw, iw=iw, col=col, nar=nar), c)
    assert np.abs(obst - so).max() <= 1e-4
    # the velocity-map files of `synthetic` (CalSurfG.f90:2559-2617) against the reference's own, byte for byte
    names = [n for n, k in (("velmap2dRc.dat", c["kRc"]), ("velmap2dRg.dat", c["kRg"]), ("velmap2dLc.dat", c["kLc"]), ("velmap2dLg.dat", c["kLg"])) if k > 0]
    for n in names:
        assert os.path.getsize(str(tmp_path / n)) > 0
    if L.ref() is not None:
        refdir = tmp_path / "ref"
        refdir.mkdir()
        cwd = os.getcwd()
        try:
            os.chdir(str(refdir))
            L.call_boundary(L.ref().synthetic_, c, synthetic=True)
        finally:
            os.chdir(cwd)
        for n in names:
            assert (tmp_path / n).read_bytes() == (refdir / n).read_bytes(), n


def test_dropin_sharded_over_engines():
    """DSA_DEVICES: the drop-in call splits its units over several engines (one per GPU; here three engines
    on GPU 0, in a fresh process because the engine pool is made once per process) -- by SOURCES since round 3, all units of a
    source on one engine, which bundles them (DSA_BUNDLE=4 here) -- and puts dsurf and the COO rows back in the reference's order;
    the last case has stations: the same sources at every period slot"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _libs as L, synth
from dsurftomo_amd import engine
lib = engine.load_library()
for kw in (dict(), dict(kRc=0, kRg=2, kLc=0, kLg=1), dict(nx=20, ny=18, nz=6, nsrc=8, nrcf=7, kRc=3, kRg=1, kLc=1, kLg=1),
           dict(nx=16, ny=15, nz=6, nsrc=7, nrcf=5, kRc=4, kRg=2, kLc=3, kLg=1, stations=True)):
    c = synth.boundary_case(**kw)
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    d = L.call_boundary(lib.dsa_calsurfg, c)
    assert o["nar"] == d["nar"], (o["nar"], d["nar"])
    assert np.abs(o["dsurf"] - d["dsurf"]).max() <= 1e-4
    assert (o["iw"] == d["iw"]).all() and (o["col"] == d["col"]).all()
    assert (o["rw"].view(np.uint32) == d["rw"].view(np.uint32)).all()
print("sharded ok")
''' % (L.ROOT, os.path.join(L.ROOT, "tests"))
    env = dict(os.environ, DSA_DEVICES="0,0,0", DSA_BUNDLE="4")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_dropin_exact_mode_from_the_environment():
    """DSA_EXACT_TIES=2: an unchanged host gets the reference's march itself behind calsurfg_; travel times,
    rays and rows of whole boundary calls -- a homogeneous model too, where every front is full of exact ties -- bit-identical"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _libs as L, synth
from dsurftomo_amd import engine
lib = engine.load_library()
for kw in (dict(), dict(kRc=0, kRg=2, kLc=0, kLg=1), dict(nx=20, ny=18, nz=6, nsrc=8, nrcf=7, kRc=3, kRg=1, kLc=1, kLg=1), dict(flat=True)):
    flat = kw.pop("flat", False)
    c = synth.boundary_case(**kw)
    if flat:
        v = np.array(c["vels"]); v[:, :, :] = v[:1, :1, :]; c["vels"] = np.asfortranarray(v)      # no lateral variation: symmetric fronts
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    d = L.call_boundary(lib.dsa_calsurfg, c)
    assert o["nar"] == d["nar"], (o["nar"], d["nar"])
    assert (o["dsurf"].view(np.uint32) == d["dsurf"].view(np.uint32)).all()
    assert (o["iw"] == d["iw"]).all() and (o["col"] == d["col"]).all()
    assert (o["rw"].view(np.uint32) == d["rw"].view(np.uint32)).all()
import ctypes as C
st = np.zeros(64); lib.dsa_dropin_engine.restype = C.c_void_p
assert lib.dsa_get_stats(C.c_void_p(lib.dsa_dropin_engine()), st.ctypes.data_as(C.c_void_p)) == 0
assert st[21] > 0, st[:26]          # DSA_STAT_EXACT_UNITS: the last call went through the literal march
print("exact ok")
''' % (L.ROOT, os.path.join(L.ROOT, "tests"))
    env = dict(os.environ, DSA_EXACT_TIES="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "exact ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_synthetic_noise_statistics(lib):
    """dsa_synthetic called directly with a noise level (the Fortran shim passes 0 and adds the host program's
    gaussian() itself): obst = t (1 + level * g), g ~ N(0, 1) from the engine's own generator"""
    c = synth.boundary_case(nx=20, ny=18, nz=5, nsrc=8, nrcf=7, kRc=3, kRg=1, kLc=1, kLg=1)
    clean = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
    noisy = L.call_boundary(lib.dsa_synthetic, c, synthetic=True, noise=0.02)
    g = (noisy / clean - 1.0) / 0.02
    assert clean.size >= 250 and np.isfinite(g).all()
    assert abs(g.mean()) < 0.25 and 0.8 < g.std() < 1.2 and np.abs(g).max() < 6.0


def test_forward_cli_on_the_taipei_directory(tmp_path):
    """python -m dsurftomo_amd.forward: the reference's input files in, residual table and matrix out"""
    out = str(tmp_path / "fw")
    r = subprocess.run([sys.executable, "-m", "dsurftomo_amd.forward", taipei.HERE, "--out", out], capture_output=True, text=True,
                       timeout=600, cwd=L.ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    tab = np.loadtxt(out + ".residual.dat")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "b_taipei.npz"))
    assert tab.shape == (2061, 3)
    assert np.abs(tab[:, 1] - z["dsurf"]).max() <= 1e-4 + 1e-6          # the text file keeps 6 decimals
    G = np.load(out + ".G.npz")
    assert int(G["rw"].size) == int(z["nar"]) and tuple(G["shape"]) == (2061, 2048)
    assert (np.diff(G["row"]) >= 0).all()


def test_dropin_empty_slots_and_sources_without_receivers(lib):
    """a period slot without sources and a source without receivers (the reference's loops simply skip them)"""
    c = synth.boundary_case(kRc=3, kRg=1, kLc=1, kLg=0, nsrc=3, nrcf=4, ragged=False)
    c["nsrcsurf1"][1] = 0                         # second Rayleigh phase period: no sources
    c["nrc1"][0, 0] = 0                           # first source of the first period: no receivers
    c["nrc1"][2, 3] = 0                           # a group-velocity source without receivers
    c["ndata"] = int(sum(c["nrc1"][s, k] for k in range(c["kmax"]) for s in range(c["nsrcsurf1"][k])))
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    d = L.call_boundary(lib.dsa_calsurfg, c)
    check_rows(o, d, c)
    assert (o["dsurf"].view(np.uint32) != d["dsurf"].view(np.uint32)).sum() == 0
    so = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)
    sd = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
    assert np.abs(so - sd).max() <= 1e-4


def test_shared_reciprocal_divisions_are_the_compilers_division():
    """dispersion_core.h: div_by and ray_core.h: divf_by -- the compiler's expansion of an IEEE division with the reciprocal's refinement shared
    between the quotients of one denominator -- against `x / d` on the device, bit for bit, over the operand ranges the kernels use them on
    (and well beyond): 0 differences in 2e8 pairs per precision, zeros, infinities and NaN among the numerators.  Outside those ranges
    (numerators within 2^53 / 2^23 of the denormal range, where v_div_scale rescales and the hand expansion does not) they may differ in the
    last bit: counted and reported, not asserted."""
    from dsurftomo_amd.engine import selfcheck_divisions
    # fp64: dispersion -- wavenumbers, layer products normalised to 1, densities: exponents -200 .. 200 is generous; fp32: rays -- coordinate
    # differences in radians (0 or >= 1e-12), travel-time differences, cell sizes 1e-6 .. 1, 6, twice a cell size in km
    n64, bad64, n32, bad32 = selfcheck_divisions(20261003, 200, [-200, 200, -200, 200, -45, 20, -25, 12])
    parity_log.add("shared-reciprocal divisions, in-contract ranges: fp64 %d pairs, %d differ; fp32 %d pairs, %d differ" % (n64, bad64, n32, bad32))
    assert n64 >= 200_000_000 and n32 >= 200_000_000
    assert bad64 == 0 and bad32 == 0
    # beyond the contract: tiny numerators
    m64, wrong64, m32, wrong32 = selfcheck_divisions(7, 20, [-1022, -960, -3, 3, -126, -100, -3, 3])
    parity_log.add("shared-reciprocal divisions, numerators below 2^-960 / 2^-100 (outside the contract): fp64 %d of %d differ, fp32 %d of %d"
           % (wrong64, m64, wrong32, m32))
