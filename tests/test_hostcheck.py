"""The product's host/device-shared logic (eikonal_core.h, source_stage.h, host_geometry.h), driven
on the CPU by tests/hostcheck.cpp, against the oracle.  No GPU involved: this covers the stencil,
the (T, tau) local solver, the serial marches, the hand-off, and the device schedule's convergence
rules (causal window, parity sub-passes, stall freeze) in emulation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _libs as L
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libhostcheck.so")


@pytest.fixture(scope="module")
def H():
    src = os.path.join(HERE, "hostcheck.cpp")
    hdr = [os.path.join(L.ROOT, "dsurftomo_amd", "csrc", n) for n in ("eikonal_core.h", "source_stage.h", "host_geometry.h")]
    hdr.append(os.path.join(HERE, "solve_node_walk_ref.h"))
    hdr.append(os.path.join(L.ROOT, "dsurftomo_amd", "csrc", "exact_march.h"))
    hdr.append(os.path.join(L.ROOT, "dsurftomo_amd", "csrc", "dispersion_core.h"))
    hdr.append(os.path.join(L.ROOT, "dsurftomo_amd", "csrc", "ray_core.h"))
    if L._stale(SO, [src] + hdr):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-msse2",
                               "-mfpmath=sse", "-shared", "-o", SO, src, "-lm"])
    h = C.CDLL(SO)
    h.hc_solve_source.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32] + [L.vp] * 7
    h.hc_gridder.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.vp]
    h.hc_fouds2_masked.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.vp, L.vp, L.vp, L.i32, L.i32]
    h.hc_fouds2_masked.restype = L.f32
    h.hc_coarse_problem.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32] + [L.vp] * 5
    h.hc_device_schedule.argtypes = [L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.f32, L.i32, L.i32, L.vp, L.vp, L.i32]
    h.hc_device_schedule.restype = C.c_long
    h.hc_solve_node_compare.argtypes = [C.c_ulonglong, C.c_long, L.vp]
    h.hc_solve_node_compare.restype = C.c_long
    h.hc_solve_regular_compare.argtypes = [C.c_ulonglong, C.c_long, L.vp]
    h.hc_solve_regular_compare.restype = C.c_long
    h.hc_regular_stats.restype = C.POINTER(C.c_long)
    h.hc_depthkernel.argtypes = [L.i32, L.i32, L.vp, L.vp, L.f32, L.i32, L.i32, L.i32, L.vp, L.i32, L.vp, L.vp, L.vp, L.vp]
    h.hc_quads_compare.argtypes = [C.c_ulonglong, C.c_long]
    h.hc_quads_compare.restype = C.c_long
    h.hc_trace_ray_lanes.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.f32, L.vp, L.vp, L.vp, L.i32]
    h.hc_exact_solve.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32, L.i32, L.i32] + [L.vp] * 5
    return h


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_solve_node_equals_the_step_by_step_form(H):
    """the product's solve_node (one evaluation body in a loop, operand-selected quadratic) == round 1's walk around the literal
    fouds2, bit for bit, on 2e7 random neighbourhoods"""
    stat = np.zeros(8, np.int64)
    bad = H.hc_solve_node_compare(20261002, 20_000_000, L.ptr(stat))
    assert bad == 0
    assert stat[1] > 1e5 and stat[2] > 1e6 and stat[3] > 1e5, stat      # the walk really takes one, two, three and more neighbours
    assert stat[5] > 100 and stat[6] > 20, stat                        # the tie detector (same (T, tau) with it) met ties, some with influence


def test_regular_form_of_the_walk_equals_solve_node(H):
    """solve_regular (round 4: the walk written out for neighbourhoods with four near neighbours, nothing pinned, tau = T -- what the bundle
    kernel's member bodies evaluate) gives solve_node's (T, tau) bit for bit wherever it says ok: 2e7 random regular neighbourhoods, and
    every evaluation of whole emulated solves on three media, where it must also cover nearly all of them"""
    stat = np.zeros(8, np.int64)
    assert H.hc_solve_regular_compare(20261003, 20_000_000, L.ptr(stat)) == 0
    assert stat[0] > 5e6, stat                                         # (the random cases are adversarial: it takes under half of them)
    assert stat[2] > 1e4, stat                                         # ... ties among them: its tie flag = solve_node_t<true>'s probe (counted in the return value)
    nx = 35
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    N = g.nnx
    sx, sz = synth.sources(nx, 8)
    rs = H.hc_regular_stats()
    for kind, src in (("rough", 1), ("checker4", 5), ("smooth", 2)):
        for k in range(8):
            rs[k] = 0
        pv = synth.medium(nx, kind)
        T = np.zeros((N, N), np.float32); tau = np.zeros((N, N), np.float32); slow = np.zeros((N, N), np.float32)
        ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32)
        assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[src], sz[src], L.ptr(T),
                                   L.ptr(tau), L.ptr(slow), L.ptr(ris), L.ptr(geom)) == 0
        out = np.zeros(4, np.int64); cyc = np.zeros(4, np.int32)
        assert H.hc_device_schedule(N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2],
                                    np.float32(0.6 * geom[3]), 1, 20000, L.ptr(out), L.ptr(cyc), 4) == 0
        evals, regular, ok, differ, noncausal = (int(rs[k]) for k in range(5))
        print("%s: %d evaluations, %d regular neighbourhoods, %d taken by solve_regular (%.2f %%), %d of them non-causal, %d differ"
              % (kind, evals, regular, ok, 100.0 * ok / evals, noncausal, differ))
        assert differ == 0
        assert ok > 0.93 * evals, (kind, evals, regular, ok)


def test_stencil_bitwise_against_oracle(H):
    """fouds2 on random alive masks over a real travel-time field"""
    nx, gd = 18, 8
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, "smooth")
    veln = L.o_gridder(g, pv)
    N = g.nnx
    sx = np.float32(g.gox + np.float32(40.3) * g.dnx)
    sz = np.float32(g.goz + np.float32(71.6) * g.dnz)
    T = L.o_solve(g, pv, veln, sx, sz)["T"]
    u = synth.LCG(5).uniform(N * N + 4000)
    alive = (u[:N * N] < 0.6).astype(np.uint8).reshape(N, N)
    pick = (u[N * N:] * (N * N)).astype(int)
    O = L.oracle()
    nmis = 0
    for p in pick[:2000]:
        ix, iz = p // N + 1, p % N + 1
        a = np.float32(O.dso_fouds2_masked(N, N, N, g.gox, g.dnx, g.dnz, g.earth, L.ptr(veln), L.ptr(T), L.ptr(alive), iz, ix))
        b = np.float32(H.hc_fouds2_masked(N, N, g.gox, g.dnx, g.dnz, g.earth, L.ptr(veln), L.ptr(T), L.ptr(alive), iz, ix))
        # the oracle returns 0 when no neighbour is alive, the product +inf
        if not np.isfinite(b):
            continue
        nmis += int(a.view(np.uint32) != b.view(np.uint32))
    assert nmis == 0


def test_trial_value_behind_a_corner_accepted_out_of_key_order(H):
    """Round 6 (tools/tie_fuzz.py, sources on node lines up to the grid's edge): a node of the refined box's narrow band whose near neighbour is a
    corner of the source cell accepted under a raised key (third, at 0.0092 s) while the node behind that corner (0.0050 s) was accepted sixth --
    the reference's trial value is the FIRST-order one fouds2 wrote at accept 3 (CalSurfG.f90:431-485: only near neighbours are evaluated again);
    the hand-off took the alive set at the exit and got the second-order value, 3 ulps lower, 4.6e-5 s downstream.  Snapshot (trial values
    included) and coarse field against the oracle, bit for bit."""
    nx, gd = 67, 8
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, "checker", 9)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    sx = np.float32(g.gox + np.float32(57.881752) * g.dnx)
    sz = np.float32(g.goz + np.float32(511.79425) * g.dnz)
    o = L.o_solve(g, pv, veln, sx, sz)
    T = np.zeros((N, N), np.float32); Tr = np.zeros(129 * 129, np.float32); Sr = np.zeros(129 * 129, np.int32)
    it = np.zeros((N, N), np.float32); is_ = np.zeros((N, N), np.int32); box = np.zeros(6, np.int32); st = np.zeros(4, np.int64)
    assert H.hc_solve_source(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd, L.ptr(pv), sx, sz, L.ptr(T), L.ptr(Tr),
                             L.ptr(Sr), L.ptr(it), L.ptr(is_), L.ptr(box), L.ptr(st)) == 0
    n = box[4] * box[5]
    Trh = Tr[:n].reshape(box[4], box[5]); Srh = Sr[:n].reshape(box[4], box[5])
    assert (np.sign(o["Sr"]).clip(-1, 1) != np.sign(Srh).clip(-1, 1)).sum() == 0
    known = o["Sr"] >= 0                                        # alive nodes and the narrow band's trial values
    assert (bits(Trh[known]) != bits(o["Tr"][known])).sum() == 0
    assert (bits(T) != bits(o["T"])).sum() == 0


@pytest.mark.parametrize("nx,kind,gd", [(18, "homog", 8), (18, "smooth", 8), (18, "smooth", 5), (35, "checker4", 8)])
def test_pipeline_against_oracle(H, nx, kind, gd):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, kind)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    vh = np.zeros((N, N), np.float32)
    H.hc_gridder(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd, L.ptr(pv), L.ptr(vh))
    assert (bits(vh) != bits(veln)).sum() == 0
    frac = [(0.43 * N + 0.3, 0.61 * N + 0.6), (1.4, N / 2 + 0.2), (N - 2.5, N - 3.3), (N - 1.0, N - 1.0), (0.0, 0.0), (N * 0.7, N * 0.2),
            (N - 2.483, 0.58 * N + 0.3)]       # out-of-order start of the coarse tree (found by tests/tools/fuzz_parity.py)
    worst, nbad = 0.0, 0
    for fx, fz in frac:
        sx = np.float32(g.gox + np.float32(fx) * g.dnx)
        sz = np.float32(g.goz + np.float32(fz) * g.dnz)
        o = L.o_solve(g, pv, veln, sx, sz)
        T = np.zeros((N, N), np.float32); Tr = np.zeros(129 * 129, np.float32); Sr = np.zeros(129 * 129, np.int32)
        it = np.zeros((N, N), np.float32); is_ = np.zeros((N, N), np.int32); box = np.zeros(6, np.int32); st = np.zeros(4, np.int64)
        assert H.hc_solve_source(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd, L.ptr(pv), sx, sz, L.ptr(T), L.ptr(Tr),
                                 L.ptr(Sr), L.ptr(it), L.ptr(is_), L.ptr(box), L.ptr(st)) == 0
        assert st[1] == 0, "serial march guard fired"
        n = box[4] * box[5]
        cls_o = np.sign(o["Sr"]).clip(-1, 1)
        cls_h = np.sign(Sr[:n].reshape(box[4], box[5])).clip(-1, 1)
        assert (cls_o != cls_h).sum() <= 4            # exact ties only (symmetric media)
        assert (np.sign(o["inj_s"]).clip(-1, 1) != np.sign(is_).clip(-1, 1)).sum() <= 2
        worst = max(worst, float(np.abs(T - o["T"]).max()))
        nbad += int((bits(T) != bits(o["T"])).sum())
    assert worst <= 1e-4 and nbad <= 0.002 * len(frac) * N * N


def test_device_schedule_converges_and_is_schedule_independent(H):
    """emulated device rounds (both evaluation modes, several windows) reach one and the same field"""
    nx = 35
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    N = g.nnx
    pv = synth.medium(nx, "rough")
    sx, sz = synth.sources(nx, 8)
    fields = []
    for mode, wc in ((1, 3.0), (0, 3.0), (1, 1.0), (1, 8.0)):
        T = np.zeros((N, N), np.float32); tau = np.zeros((N, N), np.float32); slow = np.zeros((N, N), np.float32)
        ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32)
        assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[3], sz[3], L.ptr(T), L.ptr(tau),
                                   L.ptr(slow), L.ptr(ris), L.ptr(geom)) == 0
        out = np.zeros(4, np.int64); cyc = np.zeros(4, np.int32)
        rc = H.hc_device_schedule(N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2],
                                  np.float32(wc * geom[3]), mode, 20000, L.ptr(out), L.ptr(cyc), 4)
        assert rc == 0, "no convergence"
        assert out[1] / (N * N) < 12.0, "evaluations per node out of the expected range"
        fields.append(np.abs(T))
    for f in fields[1:]:
        assert (bits(f) != bits(fields[0])).sum() == 0


def test_dependency_pruning_is_exact(H):
    """activating only the dependents that can be affected gives the same (T, tau), with far fewer evaluations"""
    nx = 35
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    N = g.nnx
    sx, sz = synth.sources(nx, 8)
    for kind, src in (("rough", 1), ("checker4", 5), ("homog", 2)):
        pv = synth.medium(nx, kind)
        res = []
        for prune in (0, 1):
            C.c_int.in_dll(H, "g_prune").value = prune
            T = np.zeros((N, N), np.float32); tau = np.zeros((N, N), np.float32); slow = np.zeros((N, N), np.float32)
            ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32)
            assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[src], sz[src], L.ptr(T),
                                       L.ptr(tau), L.ptr(slow), L.ptr(ris), L.ptr(geom)) == 0
            out = np.zeros(4, np.int64); cyc = np.zeros(4, np.int32)
            assert H.hc_device_schedule(N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2],
                                        np.float32(3.0 * geom[3]), 1, 20000, L.ptr(out), L.ptr(cyc), 4) == 0
            res.append((np.abs(T), np.abs(tau), out[1] / (N * N)))
        C.c_int.in_dll(H, "g_prune").value = 1
        assert (bits(res[0][0]) != bits(res[1][0])).sum() == 0 and (bits(res[0][1]) != bits(res[1][1])).sum() == 0
        assert res[1][2] < 0.6 * res[0][2]


@pytest.mark.parametrize("nx,kind,gd,lcap", [(18, "homog", 8, 4096), (18, "smooth", 8, 64), (18, "smooth", 5, 4096), (35, "checker4", 8, 256),
                                             (35, "rough", 8, 100), (35, "homog", 8, 4096)])
def test_exact_march_is_the_oracle_bit_for_bit(H, nx, kind, gd, lcap):
    """the exact mode's march (csrc/exact_march.h, what k_exact runs) against the oracle's Fast Marching: refined snapshot, statuses and
    the whole coarse field, every bit -- exact ties included (homogeneous and checkerboard media), with the tree split between its
    two storage parts at different depths"""
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, kind)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    frac = [(0.43 * N + 0.3, 0.61 * N + 0.6), (1.4, N / 2 + 0.2), (N - 2.5, N - 3.3), (N - 1.0, N - 1.0), (0.0, 0.0), (N * 0.7, N * 0.2),
            (N - 2.483, 0.58 * N + 0.3), (0.5 * N, 0.5 * N), (37.0, 52.0)]
    for fx, fz in frac:
        sx = np.float32(g.gox + np.float32(fx) * g.dnx)
        sz = np.float32(g.goz + np.float32(fz) * g.dnz)
        o = L.o_solve(g, pv, veln, sx, sz)
        T = np.zeros((N, N), np.float32); Tr = np.zeros(129 * 129, np.float32); Sr = np.zeros(129 * 129, np.int32)
        box = np.zeros(6, np.int32); st = np.zeros(4, np.int64)
        assert H.hc_exact_solve(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd, L.ptr(pv), sx, sz, lcap, 16 * N + 4096,
                                L.ptr(T), L.ptr(Tr), L.ptr(Sr), L.ptr(box), L.ptr(st)) == 0
        n = box[4] * box[5]
        cls_o = np.sign(o["Sr"]).clip(-1, 1)
        cls_h = Sr[:n].reshape(box[4], box[5])
        assert (cls_o != cls_h).sum() == 0
        alive = cls_o == 0
        assert (bits(Tr[:n].reshape(box[4], box[5])[alive]) != bits(o["Tr"][alive])).sum() == 0
        assert (bits(T) != bits(o["T"])).sum() == 0, (fx, fz)
        assert st[1] >= N * N - 300


def test_quadrant_form_of_the_stencil_equals_fouds2(H):
    """what the sixteen lanes of the exact mode evaluate (one quadrant each, minimum over four) == fouds2, bit for bit, on 5e6 random neighbourhoods"""
    assert H.hc_quads_compare(20261003, 5_000_000) == 0


@pytest.mark.parametrize("iwave,igr", [(2, 0), (2, 1), (1, 0), (1, 1)])
def test_dispersion_state_machine_equals_the_oracle(H, iwave, igr):
    """the product's dispersion code (csrc/dispersion_core.h: the root search unrolled into a state machine, what k_dispersion runs) on
    the CPU against the oracle's nested loops: phase / group velocities and all three depth kernels, every bit -- including a column
    with a low-velocity zone and one without any contrast"""
    c = synth.boundary_case(nx=7, ny=6, nz=7, kRc=4, kRg=3, kLc=3, kLg=2, nsrc=2, nrcf=2)
    v = np.array(c["vels"])
    v[2, 3, 2] *= 0.8; v[2, 3, 3] *= 0.75          # a low-velocity zone
    v[4, 1, :] = v[4, 1, 0]                          # a half space in disguise
    vel = np.ascontiguousarray(np.asfortranarray(v.astype(np.float32)).T)       # (nz, ny, nx)
    t = {(2, 0): c["tRc"], (2, 1): c["tRg"], (1, 0): c["tLc"], (1, 1): c["tLg"]}[(iwave, igr)]
    want = L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), iwave, igr, t)
    nz, ny, nx = vel.shape
    ncol, kmax = nx * ny, len(t)
    pv = np.zeros((kmax, ncol)); sen = [np.zeros((nz, kmax, ncol)) for _ in range(3)]
    tt = np.ascontiguousarray(t, np.float64)
    assert H.hc_depthkernel(ncol, nz, L.ptr(vel.reshape(nz, ncol)), L.ptr(np.ascontiguousarray(c["depz"], np.float32)), float(c["minthk"]), iwave, igr, kmax,
                            L.ptr(tt), 1, L.ptr(pv), L.ptr(sen[0]), L.ptr(sen[1]), L.ptr(sen[2])) == 0
    got = (pv, *sen)
    for a, b in zip(got, want):
        assert (np.ascontiguousarray(a).view(np.uint64) != np.ascontiguousarray(b).view(np.uint64)).sum() == 0


@pytest.mark.parametrize("nx,kind,gd", [(18, "smooth", 8), (18, "rough", 5), (35, "checker4", 8)])
def test_ray_trace_one_lane_and_four_lanes_against_oracle(H, nx, kind, gd):
    """trace_ray (ray_core.h) on the oracle's fields: the vertex sums (the reference's fdm, CalSurfG.f90:1970-2271), the clamp flag and the
    step count -- with one lane per ray and with four lanes sharing a ray (PatchAcc<4>: a slab row has one owner), bit for bit"""
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, kind)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    r = synth.LCG(11 + nx)
    nrays = 0
    for fx, fz in [(0.43 * N + 0.3, 0.61 * N + 0.6), (1.4, N / 2 + 0.2), (N - 2.5, N - 3.3)]:
        sx = np.float32(g.gox + np.float32(fx) * g.dnx)
        sz = np.float32(g.goz + np.float32(fz) * g.dnz)
        o = L.o_solve(g, pv, veln, sx, sz)
        u = r.uniform(16)
        for k in range(8):
            rx = np.float32(g.gox + np.float32(0.3 + u[2 * k] * (N - 1.6)) * g.dnx)
            rz = np.float32(g.goz + np.float32(0.3 + u[2 * k + 1] * (N - 1.6)) * g.dnz)
            ref, rb, ns = L.o_rpaths(g, o, veln, sx, sz, rx, rz)
            out = []
            for lanes in (1, 4):
                fdm = np.zeros((g.nvx + 2, g.nvz + 2), np.float32)
                fl, st = L.i32(0), L.i32(0)
                rc = H.hc_trace_ray_lanes(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd, L.ptr(veln), L.ptr(o["T"]),
                                          L.ptr(np.ascontiguousarray(o["Tr"])), L.ptr(np.ascontiguousarray(o["Sr"])), sx, sz, rx, rz,
                                          L.ptr(fdm), C.byref(fl), C.byref(st), lanes)
                assert rc == 0
                out.append((fdm, fl.value, st.value))
            assert out[0][1:] == out[1][1:], "four lanes per ray: clamp flag / step count differ from one lane's"
            assert out[0][1] == (rb & 1), "clamp flag differs from the oracle's"
            assert (bits(out[0][0]) != bits(out[1][0])).sum() == 0, "four lanes per ray differ from one"
            assert (bits(out[0][0]) != bits(ref)).sum() == 0 and out[0][2] == ns
            nrays += 1
    assert nrays == 24
