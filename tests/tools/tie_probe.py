import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd.engine import Engine
nx, kind = int(sys.argv[1]), sys.argv[2]
e = Engine(0)
g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
pv = synth.medium(nx, kind, 0); veln = L.o_gridder(g, pv); N = g.nnx
sx = np.float32(g.gox + np.float32(0.37 * (N - 1) + 0.3) * g.dnx); sz = np.float32(g.goz + np.float32(0.58 * (N - 1) + 0.6) * g.dnz)
o = L.o_solve(g, pv, veln, sx, sz)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.traveltimes([0], [sx], [sz], [0], [], [])
T = e.field(0); tau = e.debug_field(0, 1)
d = T - o["T"]
isx, isz = int(0.37 * (N - 1) + 0.3), int(0.58 * (N - 1) + 0.6)
print("source node (0-based ix, iz):", isx, isz)
idx = np.argsort(-np.abs(d).ravel())[:12]
for k in idx:
    ix, iz = divmod(int(k), N)
    print("ix %4d iz %4d  d %.3g  T %.6f oracle %.6f tau %.6f | dx %d dz %d" % (ix, iz, d[ix, iz], T[ix, iz], o["T"][ix, iz], abs(tau[ix, iz]), ix - isx, iz - isz))
# earliest differing nodes (smallest oracle T among differing)
diff = np.argwhere(T.view(np.uint32) != o["T"].view(np.uint32))
order = np.argsort([o["T"][i, j] for i, j in diff])[:10]
print("earliest differing nodes:")
for k in order:
    ix, iz = diff[k]
    print("ix %4d iz %4d  d %.3g  T %.7f oracle %.7f tau %.7f | dx %d dz %d" % (ix, iz, d[ix, iz], T[ix, iz], o["T"][ix, iz], abs(tau[ix, iz]), ix - isx, iz - isz))
print("sign of d: mean %.3g, frac positive %.3f" % (d[d != 0].mean(), (d > 0).sum() / max((d != 0).sum(), 1)))
