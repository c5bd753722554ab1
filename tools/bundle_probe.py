"""Bundles (bundle_kernel.hip) against unit-by-unit solves: same receiver times and fields bit for bit, and what they cost.
   python3 tools/bundle_probe.py check [nx] [nsrc] [nper] [kind]        small case: times and whole fields of every unit, bundle 0 vs 16 / 8 / 4
   python3 tools/bundle_probe.py time [nx] [nsrc] [nper] [kind] [sizes]  throughput, bundle sizes e.g. 0,16,8,4"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine

what = sys.argv[1] if len(sys.argv) > 1 else "check"
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 33
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 24
nper = int(sys.argv[4]) if len(sys.argv) > 4 else 16
kind = sys.argv[5] if len(sys.argv) > 5 else "smooth"
sizes = [int(v) for v in (sys.argv[6] if len(sys.argv) > 6 else "0,16,8,4").split(",")]


def medium(p):
    if kind == "mixed":
        i = np.arange(nx, dtype=np.float64)[None, :]; j = np.arange(nx, dtype=np.float64)[:, None]
        w = p / max(nper - 1, 1)
        v = (2.8 + 0.05 * p) * (1.0 + 0.10 * (1 - w) * np.sin(4 * np.pi * i / nx) * np.cos(4 * np.pi * j / nx)
                                + 0.08 * w * np.sin(6 * np.pi * i / nx + 1.0) * np.sin(2 * np.pi * j / nx + 0.5))
        return np.ascontiguousarray(v.reshape(-1), np.float64)
    return synth.medium(nx, kind, p)


e = Engine(0)
pv = np.stack([medium(p) for p in range(nper)])
u = synth.units(nx, nsrc, nper, 8 if what == "check" else 32)
n = nsrc * nper
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
ref = None; ref_fields = None
for G in sizes:
    if G < 0:      # a solo run with another causal window: is the solo answer itself schedule dependent here?
        e.set_option('window_cells', 1.0); G = 0
    else: e.set_option('window_cells', float(os.environ.get('DSA_PROBE_WINDOW', '1.25')))
    e.set_option('bundle', G)
    if os.environ.get('DSA_PROBE_BTHREADS'): e.set_option('bundle_threads', int(os.environ['DSA_PROBE_BTHREADS']))
    if os.environ.get('DSA_PROBE_BPOOL'): e.set_option('bundle_pool', int(os.environ['DSA_PROBE_BPOOL']))
    if os.environ.get('DSA_PROBE_MPL'): e.set_option('bundle_members_per_lane', int(os.environ['DSA_PROBE_MPL']))
    if os.environ.get('DSA_PROBE_BWINDOW'): e.set_option('bundle_window_cells', float(os.environ['DSA_PROBE_BWINDOW']))
    if what == "check": e.set_option('field_pool', -1)
    e.plan(**u)
    e.solve()                                   # (allocations)
    best = 1e9
    for rep in range(1 if what == "check" else 3):
        t0 = time.time(); t = e.solve(); best = min(best, time.time() - t0)
    st = e.stats()
    line = f'N={e.nnx} {kind} {n} units ({nsrc} sources x {nper}), bundle {G:2d}: {n/best:9.0f} solves/s, coarse kernel(s) {st["ms_fim_coarse"]:8.1f} ms, bundles {int(st["bundles"])} of size {int(st["bundle_size"])} ({int(st["bundled_units"])} units, {int(st["bundle_slots"])} slots), rounds max {int(st["rounds_max"])}, freezes {int(st["freezes"])}, member evals/node {st["evals_total"]/n/(e.nnx*e.nnz):.3f}'
    if ref is None: ref = t
    else:
        bad_t = ref.view(np.uint32) != t.view(np.uint32)
        nrec_u = t.size // n
        line += f', times identical={not bad_t.any()} (max |dt| {np.abs(ref - t).max():.3g}, {int(bad_t.sum())} of {t.size} times in {int(bad_t.reshape(n, nrec_u).any(axis=1).sum())} units: (period, source) {[(int(k) // nsrc, int(k) % nsrc) for k in np.nonzero(bad_t.reshape(n, nrec_u).any(axis=1))[0][:6]]})'
    if what == "check":
        F = np.stack([e.field(k) for k in range(n)])
        if ref_fields is None: ref_fields = F
        else:
            bad = F.view(np.uint32) != ref_fields.view(np.uint32)
            line += f', fields: {int(bad.sum())} of {bad.size} nodes differ (max |dT| {np.nanmax(np.where(bad, np.abs(F - ref_fields), 0)):.3g}; units {sorted(set(np.nonzero(bad)[0].tolist()))[:8]})'
    print(line, flush=True)
    if os.environ.get('DSA_PROBE_CLOCKS'):
        pt = np.array(st['phase_ticks'][:4]); print(f'    wall clock of the rounds (thread 0 of every bundle, -DDSA_BUNDLE_CLOCKS builds): pass A {100*pt[0]/pt.sum():.1f} %, even half {100*pt[1]/pt.sum():.1f} %, odd half {100*pt[2]/pt.sum():.1f} %, bookkeeping {100*pt[3]/pt.sum():.1f} %; per bundle-round {pt.sum()/100.0/max(st["bundles"],1)/max(np.median(e.unit_rounds()),1):.1f} us', flush=True)
    if os.environ.get('DSA_PROBE_ROUNDS'):
        r = e.unit_rounds(); q = np.percentile(r, [0, 10, 50, 90, 99, 100]).astype(int).tolist()
        worst = np.argsort(-r)[:4]
        print(f'    rounds per unit: min/10/50/90/99/max {q}; slowest units (period, source, rounds): {[(int(k) // nsrc, int(k) % nsrc, int(r[k])) for k in worst]}', flush=True)
