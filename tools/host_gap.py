"""Wall time of Engine.solve against the device time between its first and last event: python3 tools/host_gap.py <sources per period> (2 periods, 1025^2)"""
import sys, os, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import synth
from dsurftomo_amd.engine import Engine
NX, NSRC, NPER, NREC = 131, int(sys.argv[1]), 2, 32
u = synth.units(NX, NSRC, NPER, NREC)
pv = np.stack([synth.medium(NX, "smooth", p) for p in range(NPER)])
e = Engine(0)
e.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.plan(**u)
for k in range(4):
    t0 = time.perf_counter(); t = e.solve(); dt = 1e3 * (time.perf_counter() - t0)
    st = e.stats()
    print("solve %d: wall %.1f ms | device total %.1f (fim coarse %.1f refined %.1f stages %.1f) | wall - device %.1f ms" % (k, dt, st["ms_total"], st["ms_fim_coarse"], st["ms_fim_refined"], st["ms_stages"], dt - st["ms_total"]))
