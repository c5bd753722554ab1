import sys, os, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import synth
from dsurftomo_amd.engine import Engine
nsrc = int(sys.argv[1]); nt = int(sys.argv[2])
nx = 131
e = Engine(0)
pv = np.stack([synth.medium(nx, 'smooth', p) for p in range(2)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nsrc, 2, 32)
e.set_option('window_cells', float(sys.argv[3]) if len(sys.argv) > 3 else 1.25); e.set_option('fim_threads', nt)
e.plan(**u); t0 = time.time(); e.solve(); dt = time.time() - t0
st = e.stats(); n = 2 * nsrc
pt = np.array(st["phase_ticks"], dtype=float)
r = st['rounds_max']
print('units %d wg %d: %.1f solves/s fim %.1f ms; per round (us, thread 0, avg over units): pass A collect+masks %.2f expand+tau loads %.2f route %.2f | eval passes: loads %.2f solve %.2f store+activate %.2f | rounds %d | of expand+tau: expansion alone %.2f' %
      (n, nt, n / dt, st['ms_fim_coarse'], *(pt[:6] / n / r / 100), r, pt[7] / n / r / 100))
