"""What the tie detector costs where it finds nothing, and what it flags: the headline call (1025^2, 1000 sources x 16
periods) with exact_ties = 0 and exact_ties = 1 at several thresholds, on the smooth medium and on the checkerboard.
   python3 tools/ties_headline.py [nsrc] [media, comma separated] [thresholds, comma separated]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from dsurftomo_amd.engine import Engine

nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
media = (sys.argv[2] if len(sys.argv) > 2 else "smooth,checker").split(",")
thr = [float(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "2e-5,5e-6,2e-6,1e-6").split(",")]
NX, NPER, NREC = 131, 16, 32
e = Engine(0)
n = nsrc * NPER
for kind in media:
    u = synth.units(NX, nsrc, NPER, NREC, seed=synth.SEED + (41 if kind == "checker" else 0))
    pv = np.stack([synth.medium(NX, kind, p) for p in range(NPER)])
    e.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    e.set_option("exact_ties", 0); e.plan(**u); e.solve()
    best = 1e9
    for _ in range(3):
        t0 = time.time(); t_ref = e.solve(); best = min(best, time.time() - t0)
    st = e.stats()
    print(f"{kind} {n} units exact_ties=0: {n / best:9.0f} solves/s, coarse {st['ms_fim_coarse']:.1f} ms, total {st['ms_total']:.1f} ms", flush=True)
    for th in thr:
        e.set_option("exact_ties", 1); e.set_option("tie_threshold", th); e.plan(**u); e.solve()
        best = 1e9
        for _ in range(2):
            t0 = time.time(); t1 = e.solve(); best = min(best, time.time() - t0)
        st = e.stats()
        fl, inf = e.unit_ties()
        d = np.abs(t1.astype(np.float64) - t_ref.astype(np.float64)).reshape(n, NREC).max(axis=1)
        print(f"{kind} {n} units exact_ties=1 threshold {th:g}: {n / best:9.0f} solves/s, coarse {st['ms_fim_coarse']:.1f} ms, exact {st['ms_exact']:.1f} ms, total {st['ms_total']:.1f} ms, "
              f"flagged {int(st['tie_units'])} ({100.0 * st['tie_units'] / n:.2f} %), marched {int(((fl & 2) != 0).sum())}, largest influence {inf.max():.3g} s, "
              f"units whose times moved against exact_ties=0: {int((d > 0).sum())} (max {d.max():.3g} s)", flush=True)
    e.set_option("exact_ties", 0); e.set_option("tie_threshold", 2e-5)
