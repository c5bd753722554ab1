"""One source, all periods: fields of the bundle against the unit-by-unit solve; where they first differ.
   python3 tools/bundle_debug.py [nx] [kind] [nsrc_of_set] [source index] [G]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
kind = sys.argv[2] if len(sys.argv) > 2 else "checker"
nset = int(sys.argv[3]) if len(sys.argv) > 3 else 128
isrc = int(sys.argv[4]) if len(sys.argv) > 4 else 71
G = int(sys.argv[5]) if len(sys.argv) > 5 else 16
nper = 16
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
u = synth.units(nx, nset, nper, 32)
sel = np.arange(nper) * nset + isrc
rsel = (sel[:, None] * 32 + np.arange(32)[None, :]).reshape(-1)
uu = dict(map_index=u["map_index"][sel], scx=u["scx"][sel], scz=u["scz"][sel], nrec=u["nrec"][sel], rcx=u["rcx"][rsel], rcz=u["rcz"][rsel])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option('field_pool', -1)
res = {}
for g in (0, G, G):
    e.set_option('bundle', g); e.plan(**uu); t = e.solve()
    F = np.stack([e.field(k) for k in range(nper)])
    st = e.stats()
    print(f'bundle {g}: rounds {int(st["rounds_max"])} freezes {int(st["freezes"])} bundles {int(st["bundles"])}', flush=True)
    if g in res:
        print('  repeat identical:', np.array_equal(res[g][1].view(np.uint32), F.view(np.uint32)))
    res[g] = (t, F)
t0, F0 = res[0]; t1, F1 = res[G]
bad = F0.view(np.uint32) != F1.view(np.uint32)
print('times differing:', int((t0.view(np.uint32) != t1.view(np.uint32)).sum()), 'field nodes differing:', int(bad.sum()), 'per period:', bad.reshape(nper, -1).sum(axis=1).tolist())
for p in np.nonzero(bad.reshape(nper, -1).any(axis=1))[0]:
    idx = np.argwhere(bad[p])
    order = np.argsort(F0[p][bad[p]])
    print(f'period {p}: {len(idx)} nodes, max |dT| {np.abs(F0[p] - F1[p])[bad[p]].max():.3g}; earliest:')
    for k in order[:3]:
        ix, iz = idx[k]
        print(f'   node ix {ix} iz {iz}: solo {F0[p][ix, iz]!r} bundle {F1[p][ix, iz]!r}')
        for name, F in (("solo", F0[p]), ("bundle", F1[p])):
            print('     ', name, [[float(F[a, b]) if 0 <= a < F.shape[0] and 0 <= b < F.shape[1] else None for b in range(iz - 2, iz + 3)] for a in range(ix - 2, ix + 3)][2], '| x-line', [float(F[a, iz]) if 0 <= a < F.shape[0] else None for a in range(ix - 2, ix + 3)])
