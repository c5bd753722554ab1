"""Trip counters of a -DDSA_LEDGER build of the coarse solve (tools/isa_ledger.py uses them as dynamic weights).
   DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_ledger.so python3 tools/ledger_probe.py [nx] [units] [kind] > gpurun_out/ledger_counters.json"""
import sys, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = sys.argv[3] if len(sys.argv) > 3 else 'smooth'
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(2)])
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
u = synth.units(nx, nsrc // 2, 2, 32)
n = nsrc // 2 * 2
e.plan(**u)
e.solve()
st = e.stats()
c = e.debug_counters()
print(json.dumps({"grid": e.nnx, "medium": kind, "units": n, "rounds_max": st["rounds_max"], "evals_per_solve": st["evals_total"] / n, "changes_per_solve": st["changes_total"] / n,
                  "wave_trips_per_solve": {str(k): c[k] / n for k in range(24)}}))
