"""Scale probe: many units, optional chunk cap (diagnostic)."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = 131; nsrc = int(sys.argv[1]); nper = int(sys.argv[2]); cap = int(sys.argv[3])
e = Engine(0)
pv = np.stack([synth.medium(nx, 'smooth', p) for p in range(nper)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option('max_chunk', cap)
u = synth.units(nx, nsrc, nper, 32)
e.plan(**u)
t0 = time.time(); t = e.solve(); dt = time.time() - t0
st = e.stats(); n = nsrc * nper
print(f'units {n} chunkcap {cap}: {n/dt:.1f} solves/s chunk {st["chunk"]:.0f} fim_coarse {st["ms_fim_coarse"]:.0f} ms finite {np.isfinite(t).all()} sum {t.sum():.3f}', flush=True)
