import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd.engine import Engine
e = Engine(0)
CASES = [(131, "smooth", 3), (131, "checker", 0), (259, "checker", 1), (515, "checker", 2), (515, "smooth", 2)]
if len(sys.argv) > 1:
    CASES = [(int(a.split(":")[0]), a.split(":")[1], 0) for a in sys.argv[1:]]
for nx, kind, period in CASES:
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, kind, period)
    veln = L.o_gridder(g, pv)
    N = g.nnx
    sx = np.float32(g.gox + np.float32(0.37 * (N - 1) + 0.3) * g.dnx); sz = np.float32(g.goz + np.float32(0.58 * (N - 1) + 0.6) * g.dnz)
    t0 = time.time(); o = L.o_solve(g, pv, veln, sx, sz); t1 = time.time()
    e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    rng = synth.LCG(nx); u = rng.uniform(64)
    rx = (g.gox + (0.5 + u[0::2] * (N - 2)).astype(np.float32) * g.dnx).astype(np.float32)
    rz = (g.goz + (0.5 + u[1::2] * (N - 2)).astype(np.float32) * g.dnz).astype(np.float32)
    t = e.traveltimes([0], [sx], [sz], [32], rx, rz)
    st = e.stats()
    ref = np.array([L.o_srtimes(g, veln, o["T"], sx, sz, rx[k], rz[k]) for k in range(32)], np.float32)
    T = e.field(0)
    d = np.abs(T - o["T"])
    print("N=%d %s: oracle %.1fs device %.0f ms (rounds %d, evals/node %.2f) | receivers max %.3g | field: differing %.3f%%, max %.3g, p99.9 %.3g, Tmax %.1f" %
          (N, kind, t1 - t0, st["ms_total"], st["rounds_max"], st["evals_total"] / N / N, np.abs(t - ref).max(), 100 * (T.view(np.uint32) != o["T"].view(np.uint32)).mean(), d.max(), np.quantile(d, 0.999), o["T"].max()), flush=True)
