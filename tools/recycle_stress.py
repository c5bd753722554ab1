"""Stress of the recycled field slots: many units through very few slots, several times; every run must give the bits of one slot per unit.
   python3 tools/recycle_stress.py [nx] [units] [pools...]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
pools = [int(v) for v in sys.argv[3:]] or [8, 32, 257, 1024]
e = Engine(0)
pv = np.stack([synth.medium(nx, k, p) for p, k in enumerate(("checker4" if nx >= 35 else "rough", "smooth"))])
u = synth.units(nx, n // 2, 2, 4)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option('field_pool', -1); e.plan(**u)
t0 = time.time(); ref = e.solve(); print(f'N={e.nnx}: {n} units, one slot per unit: {n/(time.time()-t0):.0f} solves/s', flush=True)
for p in pools:
    e.set_option('field_pool', p); e.plan(**u)
    for rep in range(3):
        t0 = time.time(); t = e.solve(); dt = time.time() - t0
        st = e.stats()
        print(f'  pool {p:5d} run {rep}: {n/dt:8.0f} solves/s, slots {int(st["field_slots"])}, identical={np.array_equal(ref.view(np.uint32), t.view(np.uint32))}', flush=True)
