"""Throughput / scheduling probe for the fixed-point kernel (not a benchmark: see bench.py)."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
windows = [float(w) for w in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 4, 8, 16, 32]
kind = sys.argv[4] if len(sys.argv) > 4 else 'smooth'
threads = [int(t) for t in sys.argv[5].split(',')] if len(sys.argv) > 5 else [512]
e = Engine(0)
if os.environ.get('DSA_LDS_PAD'): e.set_option('fim_lds_pad', int(os.environ['DSA_LDS_PAD']))
nper = 2
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
nb = ((e.nnx + 7) // 8) ** 2
u = synth.units(nx, nsrc, nper, 32)
ref = None
variants = [int(v) for v in os.environ.get('DSA_FIM_VARIANTS', '-1').split(',')]      # fim_sorted values to run (-1: the default)
for wc, nt, var in [(w, t, v) for v in variants for t in threads for w in windows]:
    e.set_option('window_cells', wc); e.set_option('fim_threads', nt)
    if var >= 0: e.set_option('fim_sorted', var)
    e.plan(**u)
    t0 = time.time(); t = e.solve(); dt = time.time() - t0
    st = e.stats(); n = nsrc * nper
    same = 'first' if ref is None else f'identical={np.array_equal(ref.view(np.uint32), t.view(np.uint32))} maxdiff={np.abs(ref - t).max():.2g}'
    if ref is None:
        ref = t
        if os.environ.get('DSA_SAVE'): np.save(os.environ['DSA_SAVE'], t)
        if os.environ.get('DSA_COMPARE') and os.path.exists(os.environ['DSA_COMPARE']):
            other = np.load(os.environ['DSA_COMPARE'])
            print(f'      against {os.environ["DSA_COMPARE"]}: identical={np.array_equal(other.view(np.uint32), t.view(np.uint32))} differing={int((other.view(np.uint32) != t.view(np.uint32)).sum())} maxdiff={np.abs(other - t).max():.3g}', flush=True)
    print(f'N={e.nnx} {kind} variant {var:2d} units {n:5d} wg {nt:4d} window {wc:5.1f}: {n/dt:8.1f} solves/s | fim_coarse {st["ms_fim_coarse"]:8.1f} ms fim_ref {st["ms_fim_refined"]:7.1f} stages {st["ms_stages"]:6.1f} '
          f'rounds_max {st["rounds_max"]:6.0f} evals/node {st["evals_total"]/n/(e.nnx*e.nnz):5.2f} changes/node {st["changes_total"]/n/(e.nnx*e.nnz):5.2f} rescans {st["rescans"]:.0f} freezes {st["freezes"]:.0f} | {same}', flush=True)
    pt = np.array(st["phase_ticks"]); tot = pt[:4].sum()
    if os.environ.get('DSA_BARRIER_PRINT') and pt[4] > 0:   # DSA_BARRIER_CLOCKS build: waves' waits at the round's barriers
        nw = nt // 64
        print('      share of a wave\'s time spent at the barriers (mean over waves): after pass A %.3f | even half %.3f | odd half %.3f | round end %.3f | total %.3f'
              % tuple(list(pt[:4] / (nw * pt[4])) + [pt[:4].sum() / (nw * pt[4])]), flush=True)
    if os.environ.get('DSA_PASSA_PRINT') and tot > 0:      # DSA_PASSA_CLOCKS build: thread 0's sub-phase clocks (100 MHz), us per round
        r = max(st["rounds_max"], 1) * n * 100.0
        print('      us/round (thread 0, with a full wait at every tick): sweep+records %.2f | expand+tau loads %.2f (expand %.2f) | route %.2f | B: hood loads %.2f solve %.2f store+activate %.2f'
              % (pt[0] / r, pt[1] / r, pt[7] / r, pt[2] / r, pt[3] / r, pt[4] / r, pt[5] / r), flush=True)
    if tot > 0:
        print(f'      phase share: passA {pt[0]/tot:.2f} evalEven {pt[1]/tot:.2f} evalOdd {pt[2]/tot:.2f} roundEnd {pt[3]/tot:.2f} | us/unit {tot/n/100:.0f} | avg list {pt[4]/n/max(st["rounds_max"],1):.0f} avg ready/round {pt[5]/n/max(st["rounds_max"],1):.0f} max list {pt[6]:.0f}', flush=True)
