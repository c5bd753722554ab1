"""The census' per-unit tie statistics against the error of the fixed point (round 6): for the units of a call, the largest
tie influence, the SUM of the influences, their COUNT and the frozen cycles (one exact_ties = 0 run, tie_threshold ~ 0 so that every tie with an
influence is in the record) beside the worst receiver-time difference between the fixed point and the march (exact_ties = 2: the reference's bits).
Prints, for flag rules  max > a  OR  sum > b  (OR froze a cycle), the flagged fraction and the worst unit left alone; keeps the per-unit arrays in
gpurun_out/tie_scan_<tag>.npz for fitting a rule across media.  No oracle involved.
   python3 tools/tie_rule_scan.py [nx] [sources] [periods] [medium] [seed offset] [bundle]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 16
kind = sys.argv[4] if len(sys.argv) > 4 else "checker"
seed_off = int(sys.argv[5]) if len(sys.argv) > 5 else 41
bundle = int(sys.argv[6]) if len(sys.argv) > 6 else 1
nrec = 32
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + seed_off)
n = nsrc * nper
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option("bundle", bundle)
e.set_option("exact_ties", 2); e.plan(**u); tx = e.solve().reshape(n, nrec)
e.set_option("exact_ties", 0); e.set_option("tie_threshold", 1e-12); e.plan(**u); t0 = e.solve().reshape(n, nrec)
st = e.stats()
flags, mx = e.unit_ties()
cnt, sm, fr = e.unit_tie_sums()
d = np.abs(t0.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
tag = f"{kind}_{e.nnx}_{nsrc}x{nper}_s{seed_off}_b{bundle}"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"tie_scan_{tag}.npz"), d=d, mx=mx, sm=sm, cnt=cnt, fr=fr)
print(f"== {tag}: {n} units x {nrec} receivers, bundles of {int(st['bundle_size'])}; fixed point against the march: {int((d > 1e-4).sum())} units beyond 1e-4 s (worst {d.max():.3g} s), "
      f"{int((d > 5e-5).sum())} beyond 5e-5, {int((d > 0).sum())} not bit-identical; units with a tie that has an influence: {int((cnt > 0).sum())}; units that (or whose bundle) froze a cycle: {int((fr > 0).sum())}")
print(f"   per unit: ties with influence median {np.median(cnt):.0f} / 99 % {np.quantile(cnt, 0.99):.0f} / max {cnt.max()};  sum of influences median {np.median(sm):.3g} / 99 % {np.quantile(sm, 0.99):.3g} / max {sm.max():.3g} s;  "
      f"largest influence median {np.median(mx):.3g} / 99 % {np.quantile(mx, 0.99):.3g} / max {mx.max():.3g} s")
bad = np.nonzero(d > 5e-5)[0]
bad = bad[np.argsort(-d[bad])][:24]
print("   worst units (unit, error, largest influence, sum, count, froze):")
for k in bad:
    print(f"      {k:6d}  {d[k]:.3e}  {mx[k]:.3e}  {sm[k]:.3e}  {cnt[k]:5d}  {fr[k]}")
print("   rule: max > a OR sum > b          flagged     worst left alone    left alone beyond 1e-4 s")
for a in (2e-5, 1e-5, 5e-6):
    for b in (np.inf, 1e-4, 3e-5, 1e-5):
        fl = (mx > a) | (sm > b)
        rest = ~fl
        print(f"   a={a:7.1e} b={b:7.1e}   {100.0 * fl.mean():6.2f} %   {d[rest].max() if rest.any() else 0.0:10.4g} s   {int((d[rest] > 1e-4).sum()):5d}")
# the census' completeness: units in which it saw no tie with an influence should carry the reference's bits
clean = cnt == 0
print(f"   units without a tie that has an influence: {int(clean.sum())}; of them not bit-identical to the march: {int((d[clean] > 0).sum())} (worst {d[clean].max() if clean.any() else 0.0:.3g} s)")
# the map-level rule (round 6, engine option tie_map_strict): a map on which some unit holds a tie above a is tie-prone; there every unit with a tie that has an influence is flagged
MX, C, D = mx.reshape(nper, nsrc), cnt.reshape(nper, nsrc), d.reshape(nper, nsrc)
for a in (2e-5,):
    unit_fl = MX > a
    prone = unit_fl.any(axis=1)
    fl = unit_fl | (prone[:, None] & (C > 0))
    rest = ~fl
    print(f"   map-level rule a={a:.0e}: {int(prone.sum())} of {nper} maps tie-prone; flagged {100.0 * fl.mean():.2f} %; worst left alone {D[rest].max() if rest.any() else 0.0:.4g} s; left alone beyond 1e-4 s {int((D[rest] > 1e-4).sum())}, beyond 5e-5 {int((D[rest] > 5e-5).sum())}")
# the default mode as the engine runs it
e.set_option("exact_ties", 1); e.set_option("tie_threshold", 2e-5); e.plan(**u); e.solve()
t1 = e.solve().reshape(n, nrec)
st1 = e.stats()
fl1, _ = e.unit_ties()
marched = (fl1 & 2) != 0
d1 = np.abs(t1.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
print(f"   DEFAULT MODE (exact_ties = 1, tie_threshold 2e-5, tie_map_strict on): {n / (st1['ms_total'] / 1e3):.0f} solves/s ({st1['ms_total']:.0f} ms, march {st1['ms_exact']:.0f} ms); marched {int(marched.sum())} units ({100.0 * marched.mean():.2f} %), "
      f"{int(st1['tie_units_strict'])} of them by their map, {int(st1.get('tie_units_by_scale', 0))} by the size of their times, {int(st1['tie_prone_maps'])} maps tie-prone; marched units not bit-identical to exact_ties = 2: {int((d1[marched] > 0).sum())}; "
      f"units left to the fixed point: worst {d1[~marched].max() if (~marched).any() else 0.0:.4g} s, beyond 1e-4 s {int((d1[~marched] > 1e-4).sum())}, beyond 5e-5 {int((d1[~marched] > 5e-5).sum())}, holding a tie with an influence {int(st1['tie_units_tied'])}")
e.close()
