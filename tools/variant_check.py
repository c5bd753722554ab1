"""Compare the receiver times of kernel variants / workgroup sizes on the same units (must be bit-identical)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kind = sys.argv[3] if len(sys.argv) > 3 else 'smooth'
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(2)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nsrc, 2, 32)
ref = None
for sorted_, nt, w in ((1, 256, 0.4), (0, 256, 0.4), (0, 512, 3.0), (0, 1024, 1.0), (1, 512, 3.0), (1, 128, 0.2)):
    e.set_option('fim_sorted', sorted_); e.set_option('fim_threads', nt); e.set_option('window_cells', w)
    e.plan(**u)
    t = e.solve()
    st = e.stats()
    if ref is None:
        ref = t
    d = np.abs(ref - t)
    print('sorted %d wg %4d window %.1f: rounds %5.0f evals/node %.3f | differing receivers %d of %d, max %.3g' %
          (sorted_, nt, w, st['rounds_max'], st['evals_total'] / (2 * nsrc) / (e.nnx * e.nnz), int((ref.view(np.uint32) != t.view(np.uint32)).sum()), t.size, d.max()), flush=True)
