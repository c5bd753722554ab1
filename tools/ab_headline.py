"""Same-box A/B of engine options on the headline call (1025^2, sources x 16 periods, 32 receivers; bench.py's workload).
   python3 tools/ab_headline.py [nsrc] [medium] name:opt=val,opt=val [name:...] ...
Every configuration starts from the engine's defaults; the first one's receiver times are the reference the others are compared with.
Library variants: DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_<name>.so (tools/ab_build.sh)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from dsurftomo_amd.engine import Engine

nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
configs = sys.argv[3:] or ["default:"]
NX, NPER, NREC = int(os.environ.get("DSA_AB_NX", "131")), int(os.environ.get("DSA_AB_NPER", "16")), 32
reps = int(os.environ.get("DSA_AB_REPS", "3"))
n = nsrc * NPER
u = synth.units(NX, nsrc, NPER, NREC)
pv = np.stack([synth.medium(NX, kind, p) for p in range(NPER)])
ref = None
print(f"# lib {os.environ.get('DSA_LIB_PATH', 'default')}, {kind}, {nsrc} sources x {NPER} periods = {n} units, nx {NX}", flush=True)
for cfg in configs:
    name, _, opts = cfg.partition(":")
    e = Engine(0)
    for kv in [o for o in opts.split(",") if o]:
        k, v = kv.split("=")
        e.set_option(k, float(v))
    e.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    e.plan(**u)
    e.solve()
    best, bk = 1e9, 1e9
    for _ in range(reps):
        t0 = time.time(); t = e.solve(); dt = time.time() - t0
        st = e.stats()
        if dt < best: best, bk = dt, st["ms_fim_coarse"]
    line = (f"{name:28s} {n / best:9.0f} solves/s  step {1e3 * best:7.1f} ms  coarse kernel(s) {bk:7.1f} ms  refined {st['ms_fim_refined']:5.1f}  stages {st['ms_stages']:5.1f}  exact {st['ms_exact']:7.1f}  "
            f"evals/node {st['evals_total'] / n / (e.nnx * e.nnz):.4f}  bundles {int(st['bundles'])}x{int(st['bundle_size'])}  tie units {int(st['tie_units'])} (left {int(st.get('tie_units_left', 0))}, marched {int(st['exact_units'])}, max infl {st.get('tie_influence_max', 0):.3g})")
    if ref is None: ref = t
    else:
        bad = ref.view(np.uint32) != t.view(np.uint32)
        line += f"  | vs first: {int(bad.sum())} of {t.size} times differ (max {np.abs(ref - t).max():.3g} s)"
    print(line, flush=True)
    e.close()
