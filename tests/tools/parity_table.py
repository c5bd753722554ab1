"""Measured parity of every case the GPU tests assert, one line per (case, source): field max |dT| vs the oracle, nodes beyond
1e-4 s, nodes not bit-identical; boundary cases: receiver times and matrix entries.  The GPU tests' exception lists
(tests/test_gpu_parity.py TIE_CASES, tests/test_gpu_fullsize.py FULL) are written from this table.
    python3 tests/tools/parity_table.py [fixture] [full] [boundary]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L
import synth
import test_gpu_parity as P
import test_gpu_fullsize as F
import test_gpu_boundary as B
from dsurftomo_amd.engine import Engine
from dsurftomo_amd import engine as E

what = sys.argv[1:] or ["fixture", "full", "boundary"]
bits = P.bits
e = Engine(0)
if "fixture" in what:
    for nx, kind, gd in P.CASES:
        srcs = P.positions(nx, gd, P.FRAC)
        g, pv, veln, sols = P.oracle_case(nx, kind, gd, srcs)
        e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
        n = len(srcs)
        rcx = np.array([s[0] for s in srcs], np.float32); rcz = np.array([s[1] for s in srcs], np.float32)
        e.traveltimes(np.zeros(n, np.int32), [s[0] for s in srcs], [s[1] for s in srcs], np.full(n, 1, np.int32), np.roll(rcx, 3), np.roll(rcz, 3))
        for u, o in enumerate(sols):
            T = e.field(u)
            d = np.abs(T - o["T"])
            print("fixture nx=%d %-8s gd=%d src %2d (%s): field max %.3g  nodes>1e-4 %d  not identical %d of %d" %
                  (nx, kind, gd, u, P.FRAC[u], d.max(), int((d > 1e-4).sum()), int((bits(T) != bits(o["T"])).sum()), T.size), flush=True)
if "full" in what:
    for nx, kind, period, rtol, ftol, fdiff in F.FULL:
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
        e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        t = e.traveltimes([0], [sx], [sz], [32], rx, rz)
        ref = np.array([L.o_srtimes(g, veln, o["T"], sx, sz, rx[k], rz[k]) for k in range(32)], np.float32)
        T = e.field(0)
        d = np.abs(T - o["T"])
        print("full N=%d %-8s period %d: receivers max %.3g | field max %.3g  q99.9 %.3g  nodes>1e-4 %d (%.4f%%)  not identical %.3f%%  Tmax %.1f" %
              (N, kind, period, np.abs(t - ref).max(), d.max(), np.quantile(d, 0.999), int((d > 1e-4).sum()), 100.0 * (d > 1e-4).mean(), 100.0 * (bits(T) != bits(o["T"])).mean(), o["T"].max()), flush=True)
e.close()
if "boundary" in what:
    lib = E.load_library()
    from dsurftomo_amd import io as taipei
    cases = [(n, synth.boundary_case(**B.CASES[n])) for n in sorted(B.CASES)] + [("taipei", taipei.load())]
    for name, c in cases:
        o = L.call_boundary(L.oracle().dso_calsurfg, c)
        d = L.call_boundary(lib.dsa_calsurfg, c)
        Go, Gd = B.dense(o, c["ndata"], c["nparpi"]), B.dense(d, c["ndata"], c["nparpi"])
        differ = Go.view(np.uint32) != Gd.view(np.uint32)
        print("boundary %-8s: dsurf max %.3g (not identical %d of %d) | nar %d vs %d | entries differing %d, max %.3g" %
              (name, np.abs(o["dsurf"] - d["dsurf"]).max(), int((o["dsurf"].view(np.uint32) != d["dsurf"].view(np.uint32)).sum()), o["dsurf"].size,
               o["nar"], d["nar"], int(differ.sum()), np.abs(Go - Gd).max()), flush=True)
