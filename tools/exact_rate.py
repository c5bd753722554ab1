"""Throughput of the literal march alone (exact_ties = 2): python3 tools/exact_rate.py [nx] [units] [kind] [lds slots,..] [pool,..] [periods]
One warm-up solve, one timed solve per (lds, pool) pair; prints solves/s of the whole call and accepts/s of the march."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
units = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kind = sys.argv[3] if len(sys.argv) > 3 else 'checker'
lds = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else '0').split(',')]
pools = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else '0').split(',')]
nper = int(sys.argv[6]) if len(sys.argv) > 6 else 2
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, units // nper, nper, 32)
n = units // nper * nper
e.set_option('exact_ties', 2)
if os.environ.get('DSA_HEAP_BLOCKED'): e.set_option('exact_heap_blocked', int(os.environ['DSA_HEAP_BLOCKED']))      # 0: the tree's global part slot by slot
first = True
for l in lds:
    for p in pools:
        e.set_option('exact_lds_slots', l); e.set_option('exact_pool', p); e.plan(**u)
        if first and not os.environ.get('DSA_NO_WARMUP'): e.solve(); first = False
        t0 = time.time(); t = e.solve(); dt = time.time() - t0
        st = e.stats()
        print(f'N={e.nnx} {kind} {n} units, exact_ties=2, lds slots {l}, pool {p}: {n/dt:8.1f} solves/s (call), march {st["ms_exact"]:.0f} ms = {n/st["ms_exact"]*1e3:8.1f} solves/s = '
              f'{st["exact_pops"]/st["ms_exact"]/1e3:.1f} M accepts/s, {st["ms_exact"]*1e3/(st["exact_pops"]/n):.3f} us per accept per unit', flush=True)
        c = e.debug_counters()
        if c[:8].sum() > 0:          # probe build -DDSA_X_CLOCKS: cycles per phase of the accept step, summed over the wavefronts' first lanes
            names = ["root+coords", "indices+fetch issue", "pop", "slot fix-up + fetch wait", "paths + candidates", "votes + stores (+ sequential way)", "-"]
            steps = st["exact_pops"] / n * (n / 4.0)
            print("    cycles per step: " + ", ".join(f"{nm} {c[i]/steps:.0f}" for i, nm in enumerate(names)) + f" | total {c[:7].sum()/steps:.0f} | steps the sequential way {c[7]/steps:.3f}", flush=True)
