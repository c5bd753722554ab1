"""What the tie detector's threshold buys (engine option tie_threshold, exact_ties = 1): for the units of a call, the largest tie influence
each unit met (one run with the detector on and a threshold nothing reaches) against the error of its default-mode receiver times
(exact_ties = 0 against exact_ties = 2, which is the reference's answer bit for bit) -- flagged fraction and the worst unit left alone,
threshold by threshold.  No oracle involved.
   python3 tools/tie_threshold_scan.py [nx] [sources] [periods] [medium]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 16
kind = sys.argv[4] if len(sys.argv) > 4 else "checker"
nrec = 32
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + 41)
n = nsrc * nper
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option("exact_ties", 2); e.plan(**u); tx = e.solve().reshape(n, nrec)
e.set_option("exact_ties", 0); e.set_option("bundle", 0); e.plan(**u); t0 = e.solve().reshape(n, nrec)
e.set_option("exact_ties", 1); e.set_option("tie_threshold", 1e-12); e.plan(**u); t1 = e.solve().reshape(n, nrec)
flags, infl = e.unit_ties()
frozen = ((flags & 1) != 0) & (infl <= 0)                      # (every tie with an influence counts in this run; the frozen ones are found through their zero influence below)
d = np.abs(t0.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
print(f"N={e.nnx} {kind}: {n} units x {nrec} receivers; default mode against the march: {int((d > 1e-4).sum())} units with a receiver beyond 1e-4 s (worst {d.max():.3g} s), "
      f"{int((d > 0).sum())} units with a time not bit-identical; {int(frozen.sum())} units froze a cycle")
print("threshold (s)   flagged    worst |dt| of the units left alone   units left alone beyond 1e-4 s")
for th in (0.0, 1e-6, 2e-6, 5e-6, 1e-5, 1.5e-5, 2e-5, 3e-5, 5e-5, 1e-4):
    fl = frozen | (infl > th)                  # (threshold 0: any tie with an influence at all)
    rest = ~fl
    print(f"{th:10.1e}   {100.0 * fl.mean():6.2f} %   {d[rest].max() if rest.any() else 0.0:12.4g} s   {int((d[rest] > 1e-4).sum()):6d}")
e.close()
