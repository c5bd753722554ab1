"""Throughput of the exact mode (literal Fast Marching, one wavefront per unit) and what the tie detector flags.
   python3 tools/exact_probe.py [nx] [units] [kind] [lds slots list] [pool list]"""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = sys.argv[3] if len(sys.argv) > 3 else 'smooth'
lds = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else '2048').split(',')]
pools = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else '0').split(',')]
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(2)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nsrc // 2, 2, 32)
n = nsrc // 2 * 2
e.set_option('exact_ties', 0); e.plan(**u)
t0 = time.time(); tf = e.solve(); dt = time.time() - t0
print(f'N={e.nnx} {kind} {n} units: fixed point {n/dt:8.1f} solves/s', flush=True)
e.set_option('exact_ties', 1); e.plan(**u)
t0 = time.time(); t1 = e.solve(); dt = time.time() - t0
st = e.stats(); fl, inf = e.unit_ties()
print(f'  exact_ties=1: {n/dt:8.1f} solves/s | flagged {int(st["tie_units"])} of {n} units ({100.0*st["tie_units"]/n:.1f} %), largest influence {inf.max():.3g} s, '
      f'median of flagged {np.median(inf[fl & 1 > 0]) if (fl & 1).any() else 0:.3g} s | exact part {st["ms_exact"]:.0f} ms | receivers differing from the fixed point: '
      f'{int((t1.view(np.uint32) != tf.view(np.uint32)).sum())} of {t1.size}, max {np.abs(t1 - tf).max():.3g} s', flush=True)
e.set_option('exact_ties', 2); e.set_option('exact_lds_slots', lds[0]); e.plan(**u)
tx = e.solve()                      # the literal march for every unit: the reference to judge the thresholds by
nrec = 32
dfix = np.abs(tf.astype(np.float64) - tx.astype(np.float64)).reshape(n, nrec).max(axis=1)
print(f'  fixed point vs literal march: units with a receiver beyond 1e-4 s: {int((dfix > 1e-4).sum())} of {n}, worst {dfix.max():.3g} s', flush=True)
e.set_option('exact_ties', 1)
for thr in (() if os.environ.get('DSA_PROBE_SKIP_THRESHOLDS') else (1e-6, 5e-6, 1e-5, 2e-5, 3e-5, 5e-5, 1e-4)):
    e.set_option('tie_threshold', thr); e.plan(**u); tt = e.solve(); st = e.stats(); fl, inf = e.unit_ties()
    left = (fl & 2) == 0
    d = np.abs(tt.astype(np.float64) - tx.astype(np.float64)).reshape(n, nrec).max(axis=1)
    print(f'  tie_threshold {thr:g}: flagged {int(st["tie_units"])} of {n} ({100.0*st["tie_units"]/n:.1f} %) | units left to the fixed point: worst receiver |dt| {d[left].max() if left.any() else 0:.3g} s, '
          f'{int((d[left] > 1e-4).sum())} beyond 1e-4 s | flagged units identical to the march: {bool((d[~left] == 0).all())} | {n/ (st["ms_total"]/1e3 + 1e-9):.0f} solves/s (device time)', flush=True)
e.set_option('tie_threshold', 2e-5)
e.set_option('exact_ties', 2)
ref = None
for l in lds:
    for p in pools:
        e.set_option('exact_lds_slots', l); e.set_option('exact_pool', p); e.plan(**u)
        t0 = time.time(); t2 = e.solve(); dt = time.time() - t0
        st = e.stats()
        if ref is None: ref = t2
        print(f'  exact_ties=2 lds slots {l:5d} pool {p:5d}: {n/dt:8.1f} solves/s | exact part {st["ms_exact"]:.0f} ms = {st["exact_pops"]/st["ms_exact"]/1e3:.2f} M accepts/s, '
              f'{st["ms_exact"]*1e3/ (st["exact_pops"]/n) :.3f} us per accept per unit-slot | identical to first {np.array_equal(ref.view(np.uint32), t2.view(np.uint32))} | '
              f'vs fixed point: differing {int((t2.view(np.uint32) != tf.view(np.uint32)).sum())} of {t2.size}, max {np.abs(t2 - tf).max():.3g} s', flush=True)
