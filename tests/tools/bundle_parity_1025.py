"""Bundled and unit-by-unit solves at 1025^2 against the oracle's Fast Marching, receiver times of S sources x 16 periods (32 receivers each),
media smooth / checker / rough: max |dt|, times beyond 1e-4 s, times not bit-identical -- for the record (profiles/), nothing asserted.
    python3 tests/tools/bundle_parity_1025.py [sources]"""
import os, sys, numpy as np
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L, synth
from dsurftomo_amd.engine import Engine
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx, nper, nrec = 131, 16, 32
e = Engine(0)
g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
for kind in ("smooth", "checker", "rough"):
    pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
    u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + 21)
    n = nsrc * nper
    veln = [L.o_gridder(g, pv[p]) for p in range(nper)]

    def one(k):
        p = int(u["map_index"][k])
        o = L.o_solve(g, pv[p], veln[p], u["scx"][k], u["scz"][k])
        return np.array([L.o_srtimes(g, veln[p], o["T"], u["scx"][k], u["scz"][k], u["rcx"][k * nrec + r], u["rcz"][k * nrec + r]) for r in range(nrec)], np.float32)

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        ref = np.stack(list(ex.map(one, range(n))))
    e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    for G in (0, 16):
        e.set_option("bundle", G)
        t = e.traveltimes(**u).reshape(n, nrec)
        d = np.abs(t.astype(np.float64) - ref.astype(np.float64))
        print("N=1025 %-7s %3d units x %d receivers, %s: max |dt| %.3g s, beyond 1e-4 s %d of %d, not bit-identical %d" %
              (kind, n, nrec, "bundles of 16 " if G else "unit by unit  ", d.max(), int((d > 1e-4).sum()), d.size, int((t.view(np.uint32) != ref.view(np.uint32)).sum())), flush=True)
