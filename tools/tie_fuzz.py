"""Fresh-seed fuzz of the DEFAULT mode's tolerance (round 6): calls on tie-prone and smooth media at several grid sizes, every receiver
time of the default mode (exact_ties = 1: fixed point, census, the march for the flagged units -- the per-unit rule and the map-level rule) against
exact_ties = 2 (the reference's Fast Marching replayed on the device: pinned bit for bit by the GPU tests).  Per call: units, marched, the units left to
the fixed point that end beyond 1e-4 s (the escapees), their worst time; the units in which the census saw no tie with an influence and that still
differ from the march (what the census does not see).  No oracle involved.
   python3 tools/tie_fuzz.py [first seed offset] [calls per medium]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import synth
from dsurftomo_amd.engine import Engine

import fuzz_sources; fuzz_sources.install()      # (DSA_FUZZ_INNER, DSA_FUZZ_SNAP: where the sources go)
DICING = int(os.environ.get("DSA_FUZZ_DICING", "8"))
VSCALE = float(os.environ.get("DSA_FUZZ_VSCALE", "1"))      # every velocity times this: 0.8 puts the far units of the 1025^2 calls beyond 64 s -- the edge of the default mode's envelope (tie_scale_guard)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nrec = 32
#          nx   sources periods medium
CONFIGS = [(131, 1000, 16, "checker"), (131, 1000, 16, "rough"), (131, 500, 16, "wild"), (67, 1000, 16, "checker"), (67, 1000, 16, "rough"), (35, 1000, 16, "checker4"),
           (35, 1000, 16, "rough"), (131, 1000, 16, "smooth"), (67, 1000, 16, "smooth"), (35, 1000, 16, "smooth")]
if os.environ.get("DSA_FUZZ_SMALL"):      # (the sizes DSurfTomo's users run: 121^2 .. 193^2)
    CONFIGS = [(18, 1000, 16, "smooth"), (18, 1000, 16, "checker4"), (18, 1000, 16, "rough"), (27, 1000, 16, "smooth"), (27, 1000, 16, "checker4"), (27, 1000, 16, "homog"), (18, 1000, 16, "homog")]
e = Engine(0)
tot = {}
t_start = time.time()
only = [int(v) for v in os.environ.get("DSA_FUZZ_ONLY", "").split(",") if v]      # (indices into CONFIGS: one call again, with the same seed)
for ci, (nx, nsrc, nper, kind) in enumerate(CONFIGS):
    if only and ci not in only:
        continue
    for call in range(ncalls):
        seed = seed0 + 17 * ci + call
        pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)]) * VSCALE
        u = synth.units(nx, nsrc, nper, nrec, gd=DICING, seed=synth.SEED + seed)
        n = nsrc * nper
        e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=DICING)
        e.set_option("exact_ties", 2); e.plan(**u); tx = e.solve().reshape(n, nrec)
        e.set_option("exact_ties", 1); e.plan(**u); t1 = e.solve().reshape(n, nrec)
        st = e.stats()
        fl, mx = e.unit_ties()
        cnt, sm, fr = e.unit_tie_sums()
        marched = (fl & 2) != 0
        d = np.abs(t1.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
        left = ~marched
        clean = left & (cnt == 0) & (fr == 0)
        rec = dict(units=n, marched=int(marched.sum()), marched_differ=int((d[marched] > 0).sum()), left=int(left.sum()), left_tied=int((left & (cnt > 0)).sum()),
                   escapees=int((d[left] > 1e-4).sum()), beyond_5e5=int((d[left] > 5e-5).sum()), worst=float(d[left].max()) if left.any() else 0.0,
                   clean=int(clean.sum()), clean_differ=int((d[clean] > 0).sum()), clean_worst=float(d[clean].max()) if clean.any() else 0.0,
                   froze=int((fr > 0).sum()), left_froze_differ=int((d[left & (fr > 0)] > 0).sum()))
        print(f"N={e.nnx:5d} {kind:8s} seed+{seed}: {n} units, {int(st['tie_prone_maps'])}/{nper} maps tie-prone, marched {rec['marched']} ({int(st.get('tie_units_by_scale', 0))} by the size of their times; refined boxes replayed at the hand-off {int(st.get('handoffs_replayed', 0))}; not bit-identical to exact_ties=2: {rec['marched_differ']}); "
              f"left to the fixed point {rec['left']} (holding a tie with an influence {rec['left_tied']}): beyond 1e-4 s {rec['escapees']}, beyond 5e-5 {rec['beyond_5e5']}, worst {rec['worst']:.3g} s; "
              f"no tie seen {rec['clean']}, of them not bit-identical {rec['clean_differ']} (worst {rec['clean_worst']:.3g} s); units whose bundle froze a cycle {rec['froze']} (left alone and not bit-identical: {rec['left_froze_differ']}) | {n / (st['ms_total'] / 1e3):.0f} solves/s", flush=True)
        grp = "smooth" if kind in ("smooth", "homog") else "tie-prone"
        t = tot.setdefault(grp, dict(units=0, marched=0, marched_differ=0, left=0, left_tied=0, escapees=0, beyond_5e5=0, worst=0.0, clean=0, clean_differ=0, clean_worst=0.0, froze=0, left_froze_differ=0))
        for k2 in rec:
            t[k2] = max(t[k2], rec[k2]) if k2 in ("worst", "clean_worst") else t[k2] + rec[k2]
for grp, t in tot.items():
    print(f"TOTAL {grp}: {t['units']} units x {nrec} receivers; marched {t['marched']} (differ from exact_ties=2: {t['marched_differ']}); left to the fixed point {t['left']} ({t['left_tied']} of them hold a tie with an influence): "
          f"beyond 1e-4 s {t['escapees']}, beyond 5e-5 {t['beyond_5e5']}, worst {t['worst']:.3g} s; units without a tie seen {t['clean']}, not bit-identical {t['clean_differ']} (worst {t['clean_worst']:.3g} s)")
print(f"({time.time() - t_start:.0f} s)")
e.close()
