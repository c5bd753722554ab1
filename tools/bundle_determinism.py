"""Determinism of the bundled solve: the same call again and again, every receiver time compared bit for bit with the first run.
   python3 tools/bundle_determinism.py [nx] [nsrc] [nper] [kind] [reps]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 16
kind = sys.argv[4] if len(sys.argv) > 4 else "smooth"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
e = Engine(0)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, np.stack([synth.medium(nx, kind, p) for p in range(nper)]))
e.plan(**synth.units(nx, nsrc, nper, 32))
if os.environ.get('DSA_BUNDLE'): e.set_option('bundle', int(os.environ['DSA_BUNDLE'])); e.plan(**synth.units(nx, nsrc, nper, 32))
ref = e.solve(); st = e.stats(); bad = 0; worst = 0.0; runs = 0
for r in range(reps):
    t = e.solve()
    nb = int((t.view(np.uint32) != ref.view(np.uint32)).sum())
    bad += nb; runs += nb > 0; worst = max(worst, float(np.abs(t - ref).max()))
print(f'N={e.nnx} {kind}: {nsrc * nper} units in {int(st["bundles"])} bundles of {int(st["bundle_size"])}, {reps} repeats, {ref.size} receiver times each: {bad} differing from the first run (in {runs} of the runs; largest |dt| {worst:.3g} s)')
