"""The march on whole fields against the march in pooled tiles (engine option exact_tiles; csrc/exact_kernel.hip xg_tile_*), times-only calls:
   python3 tools/exact_tiles_probe.py [nx] [nsrc] [nper] [kind] [modes, e.g. -1,1] [tile cap]
Rates by the engine's HIP events over the whole call; receiver times compared bit for bit with the first mode's."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from dsurftomo_amd.engine import Engine

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 515
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 24
kind = sys.argv[4] if len(sys.argv) > 4 else "checker"
modes = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "-1,1").split(",")]
cap = int(sys.argv[6]) if len(sys.argv) > 6 else 0
n = nsrc * nper
u = synth.units(nx, nsrc, nper, 32, seed=synth.SEED + 47)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
e = Engine(0)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.set_option("exact_ties", 2)
e.set_option("exact_tile_cap", cap)
ref = None
for md in modes:
    e.set_option("exact_tiles", md)
    e.plan(**u)
    t = e.solve()
    st = e.stats()
    line = (f"N={e.nnx} {kind} {n} units exact_ties=2 exact_tiles={md:2d}: {n / (st['ms_total'] / 1e3):8.1f} solves/s, march {st['ms_exact']:.0f} ms = {st['exact_pops'] / st['ms_exact'] / 1e3:.0f} M accepts/s, "
            f"{int(st['exact_pool'])} units side by side" + (f", {int(st['exact_tiles'])} tiles per unit" if st['exact_tiles'] else ", whole fields"))
    if ref is None: ref = t
    else: line += f" | {int((ref.view(np.uint32) != t.view(np.uint32)).sum())} of {t.size} receiver times differ from the first mode"
    print(line, flush=True)
