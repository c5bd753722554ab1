"""The reference's own tie sensitivity on the random sources of tests/tools/fuzz_parity.py (1025^2, rough medium, seed 5): for how many
fields does preferring the other child on equal keys in the reference's heap (DSO_TIE_POLICY=1, see tests/tools/tie_sensitivity.py) move
some node by more than 1e-4 s?  To be read next to the GPU engine's figures for the same sources (profiles/r02_fuzz_parity.log).  CPU only."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nx, kind, nsrc, seed = 131, sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] != "--solve" else "rough", 48, 5
if len(sys.argv) > 1 and sys.argv[1] == "--solve":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L, synth
    kind = sys.argv[3]
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, kind); veln = L.o_gridder(g, pv); N = g.nnx
    r = synth.LCG(seed * 1000 + nx)
    u = r.uniform(3 * nsrc)
    fx = u[0::3] * (N - 1); fz = u[1::3] * (N - 1); snap = u[2::3]
    fx = np.where(snap < 0.15, np.round(fx), fx); fz = np.where((snap > 0.1) & (snap < 0.25), np.round(fz), fz)
    fx = np.where(snap > 0.9, np.where(fx > N / 2, N - 1 - 0.3 * (1 - snap) * 10, 0.3 * (1 - snap) * 10), fx)
    sx = (g.gox + fx.astype(np.float32) * g.dnx).astype(np.float32); sz = (g.goz + fz.astype(np.float32) * g.dnz).astype(np.float32)
    sx = np.clip(sx, g.gox, np.float32(g.gox + np.float32(N - 1) * g.dnx)); sz = np.clip(sz, g.goz, np.float32(g.goz + np.float32(N - 1) * g.dnz))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as ex:
        T = list(ex.map(lambda k: L.o_solve(g, pv, veln, sx[k], sz[k])["T"], range(nsrc)))
    np.save(sys.argv[2], np.stack(T))
    sys.exit(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "rough"
tmp = "/tmp/tsf_%d" % os.getpid()
for pol in ("0", "1"):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--solve", "%s_%s.npy" % (tmp, pol), kind], env=dict(os.environ, DSO_TIE_POLICY=pol))
a, b = np.load(tmp + "_0.npy"), np.load(tmp + "_1.npy")
live = a.reshape(nsrc, -1).max(axis=1) > 0
d = np.abs(a - b).reshape(nsrc, -1)
over = (d > 1e-4).sum(axis=1)
print("N=1025 %s, %d random sources (%d degenerate in the reference): reference vs reference with the other tie preference: fields with a node beyond 1e-4 s: %d "
      "(max %.3g s, %.4f %% of all nodes); bit-identical fields %d; nodes not bit-identical %.3f %%" %
      (kind, nsrc, int((~live).sum()), int((over[live] > 0).sum()), d.max(), 100.0 * over[live].sum() / max(d[live].size, 1),
       int(((a.reshape(nsrc, -1).view(np.uint32) != b.reshape(nsrc, -1).view(np.uint32)).sum(axis=1)[live] == 0).sum()),
       100.0 * (a.view(np.uint32) != b.view(np.uint32)).mean()))
os.remove(tmp + "_0.npy"); os.remove(tmp + "_1.npy")
