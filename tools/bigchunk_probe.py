import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nunits_src = int(sys.argv[1]); budget = float(sys.argv[2])
e = Engine(0)
e.set_memory_budget(int(budget * 1e9))
nx = 131
pv = np.stack([synth.medium(nx, "smooth", p) for p in range(16)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nunits_src, 16, 32)
e.plan(**u)
for k in range(2):
    t0 = time.time(); t = e.solve(); dt = time.time() - t0
    st = e.stats()
    print("units %d budget %.0f GB: chunk %d, %.1f solves/s, fim %.0f ms, launches %d" % (16 * nunits_src, budget, st["chunk"], 16 * nunits_src / dt, st["ms_fim_coarse"], st["launches_fim_coarse"]), flush=True)
