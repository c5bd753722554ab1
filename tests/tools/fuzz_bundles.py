"""Fuzz the BUNDLED solve (engine option bundle; csrc/bundle_kernel.hip) against the oracle and against the unit-by-unit solve: random sources
(some exactly on nodes, some hugging an edge) x 4 maps of different media on small grids, forced bundles of 4 / 8, whole fields of every unit.
    python tests/tools/fuzz_bundles.py [nsrc] [seed] [nx ...]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd.engine import Engine
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sizes = [int(a) for a in sys.argv[3:]] or [18, 27, 35]
KINDS = ("smooth", "rough", "checker4", "smooth", "rough", "homog", "checker", "rough")       # one medium per period: members of a bundle differ in kind
e = Engine(0)
gd = 8
for nx in sizes:
    for G, nper in ((4, 4), (8, 7)):
        g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
        N = g.nnx
        pv = np.stack([synth.medium(nx, KINDS[p], p) for p in range(nper)])
        veln = [L.o_gridder(g, pv[p]) for p in range(nper)]
        r = synth.LCG(seed * 1000 + nx + G)
        u = r.uniform(3 * nsrc)
        fx = u[0::3] * (N - 1); fz = u[1::3] * (N - 1); snap = u[2::3]
        fx = np.where(snap < 0.15, np.round(fx), fx); fz = np.where((snap > 0.1) & (snap < 0.25), np.round(fz), fz)
        fx = np.where(snap > 0.9, np.where(fx > N / 2, N - 1 - 0.3 * (1 - snap) * 10, 0.3 * (1 - snap) * 10), fx)
        sx = np.clip((g.gox + fx.astype(np.float32) * g.dnx).astype(np.float32), g.gox, np.float32(g.gox + np.float32(N - 1) * g.dnx))
        sz = np.clip((g.goz + fz.astype(np.float32) * g.dnz).astype(np.float32), g.goz, np.float32(g.goz + np.float32(N - 1) * g.dnz))
        n = nsrc * nper
        mi = np.repeat(np.arange(nper, dtype=np.int32), nsrc); SX = np.tile(sx, nper); SZ = np.tile(sz, nper)
        e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
        e.set_option("field_pool", -1)
        F = {}
        for b in (0, G):
            e.set_option("bundle", b)
            e.traveltimes(mi, SX, SZ, np.zeros(n, np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32))
            st = e.stats()
            F[b] = np.stack([e.field(k) for k in range(n)])
            if b: assert st["bundles"] > 0, st
        vs_solo = int((F[G].view(np.uint32) != F[0].view(np.uint32)).sum())
        nbad = nexact = ndeg = 0; mx = 0.0; mx_solo = 0.0
        for k in range(n):
            p = int(mi[k])
            o = L.o_solve(g, pv[p], veln[p], SX[k], SZ[k])
            if o["T"].max() == 0.0: ndeg += 1; continue
            d = float(np.abs(F[G][k] - o["T"]).max()); mx = max(mx, d); mx_solo = max(mx_solo, float(np.abs(F[0][k] - o["T"]).max()))
            nexact += int((F[G][k].view(np.uint32) != o["T"].view(np.uint32)).sum() == 0)
            nbad += d > 1e-4
        print("nx %d (N=%d), %d sources x %d media in bundles of %d: fields bit-identical to the oracle %d of %d, with a node over 1e-4 s: %d (max %.3g s; unit by unit: max %.3g s); "
              "nodes differing from the unit-by-unit solve: %d of %d" % (nx, N, nsrc, nper, G, nexact, n - ndeg, nbad, mx, mx_solo, vs_solo, F[0].size), flush=True)
