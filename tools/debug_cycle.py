"""Diagnostic: run the failing parity case and print the nodes still queued when the round cap hits."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine, EngineError
from test_gpu_parity import positions, FRAC
nx, kind, gd = 35, 'rough', 8
e = Engine(0)
pv = synth.medium(nx, kind)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
srcs = positions(nx, gd, FRAC)
n = len(srcs)
failed = None
for attempt in range(40):
    try:
        e.traveltimes(np.zeros(n, np.int32), [s[0] for s in srcs], [s[1] for s in srcs], np.zeros(n, np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    except EngineError as ex:
        failed = ex
        break
print('attempts', attempt + 1, 'last stats', e.stats())
if failed is not None:
    ex = failed
    print(ex)
    for u in range(n):
        tau = e.debug_field(u, 1); T = e.debug_field(u, 0)
        q = np.argwhere(np.signbit(tau) & ~np.signbit(T))
        print(f'unit {u}: queued nodes {len(q)}; unreached {(~np.isfinite(T)).sum()}')
        for (ix, iz) in q[:6]:
            sl = (slice(max(ix-2,0), ix+3), slice(max(iz-2,0), iz+3))
            print(f'   node ix={ix+1} iz={iz+1} T={T[ix,iz]:.7f} tau={abs(tau[ix,iz]):.7f}')
            print('     T   :', np.array2string(np.abs(T[sl]), precision=6, max_line_width=200).replace('\n', '\n           '))
            print('     tau :', np.array2string(np.abs(tau[sl]), precision=6, max_line_width=200).replace('\n', '\n           '))
