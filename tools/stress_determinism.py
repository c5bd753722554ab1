"""Run one small case many times; every run must give bit-identical fields (the fixed point is
schedule independent).  Prints the first differing nodes if not."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
from test_gpu_parity import positions, FRAC
nx, kind, gd = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]); reps = int(sys.argv[4])
e = Engine(0)
pv = synth.medium(nx, kind)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
srcs = positions(nx, gd, FRAC); n = len(srcs)
args = (np.zeros(n, np.int32), [s[0] for s in srcs], [s[1] for s in srcs], np.zeros(n, np.int32), np.zeros(0, np.float32), np.zeros(0, np.float32))
e.traveltimes(*args)
ref = [e.debug_field(u, 0).copy() for u in range(n)]
reft = [e.debug_field(u, 1).copy() for u in range(n)]
refr = [e.debug_field(u, 2).copy() for u in range(n)]
bad_runs = 0
for r in range(reps):
    e.traveltimes(*args)
    for u in range(n):
        T = e.debug_field(u, 0); Tr = e.debug_field(u, 2)
        d = np.argwhere(T.view(np.uint32) != ref[u].view(np.uint32))
        dr = np.argwhere(Tr.view(np.uint32) != refr[u].view(np.uint32))
        if len(d) or len(dr):
            bad_runs += 1
            tau = e.debug_field(u, 1)
            print(f'run {r} unit {u}: coarse nodes differ {len(d)}, refined differ {len(dr)}; stats {e.stats()}')
            if len(d):
                order = np.argsort([abs(ref[u][i, j]) for i, j in d])
                for k in order[:5]:
                    i, j = d[k]
                    print(f'   ix={i+1} iz={j+1}: T {T[i,j]:.7f} (first run {ref[u][i,j]:.7f}) tau {tau[i,j]:.7f} (first {reft[u][i,j]:.7f})')
            break
print(f'{kind} nx={nx} gd={gd}: {bad_runs} of {reps} runs differ from the first run')
