"""Sweep kernel options at the headline grid: python tools/sweep_probe.py nsrc 'window,threads,sorted;...'"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nsrc = int(sys.argv[1]); combos = [tuple(float(v) for v in c.split(',')) for c in sys.argv[2].split(';')]
kind = sys.argv[3] if len(sys.argv) > 3 else 'smooth'
nx = 131
e = Engine(0)
if os.environ.get('DSA_LDS_PAD'): e.set_option('fim_lds_pad', int(os.environ['DSA_LDS_PAD']))
pv = np.stack([synth.medium(nx, kind, p) for p in range(2)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nsrc, 2, 32)
ref = None
for w, nt, srt in combos:
    e.set_option('window_cells', w); e.set_option('fim_threads', int(nt)); e.set_option('fim_sorted', int(srt))
    e.plan(**u)
    t0 = time.time(); t = e.solve(); dt = time.time() - t0
    st = e.stats(); n = 2 * nsrc
    if ref is None: ref = t
    same = np.array_equal(ref.view(np.uint32), t.view(np.uint32))
    if not same: same = 'NO: %d receivers differ, max %.3g, freezes %d' % (int((ref.view(np.uint32) != t.view(np.uint32)).sum()), np.abs(ref - t).max(), st['freezes'])
    print('window %.2f wg %4d sorted %d: %7.1f solves/s | fim %.1f ms rounds %5.0f evals/node %.2f | identical %s' %
          (w, nt, srt, n / dt, st['ms_fim_coarse'], st['rounds_max'], st['evals_total'] / n / (e.nnx * e.nnz), same), flush=True)
