#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU's share: 4097^2 grid (nx = ny = 515), checkerboard +-8 % with 16-vertex squares, 512 (period, source)
units (of the 4000 x 24 / 8 = 12 000 a GPU of the 8-GPU job would take, in chunks of ~1000 = 150 GB), 32 receivers each: throughput of the
solve, and the receiver times of a sample of units against the oracle's Fast Marching (oracle solves spread over the host cores).
    python3 tests/tools/config4_probe.py [units] [sampled units] [periods]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L, synth
from dsurftomo_amd.engine import Engine

nunits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nx, nrec = 515, 32
nsrc = nunits // nper
u = synth.units(nx, nsrc, nper, nrec)
pv = np.stack([synth.medium(nx, "checker", p) for p in range(nper)])
e = Engine(0)
if os.environ.get('DSA_BUNDLE'): e.set_option('bundle', int(os.environ['DSA_BUNDLE']))
t0 = time.perf_counter()
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
e.plan(**u)
print("N = %d, %d units (%d sources x %d periods), %d receivers each; setup %.2f s" % (e.nnx, nsrc * nper, nsrc, nper, nrec, time.perf_counter() - t0), flush=True)
for k in range(2):
    t0 = time.perf_counter(); t = e.solve(); dt = time.perf_counter() - t0
    st = e.stats()
    print("solve %d: %.2f s wall = %.1f solves/s | coarse fixed-point kernel %.1f ms (%.1f solves/s, %.1f GB/s of algorithmic bytes), refined %.1f ms, other stages %.1f ms | "
          "rounds_max %d, evaluations per node %.2f, %d units per launch through %d field slots, footprint %.1f GB; bundles %d of %d members in %d slots" %
          (k, dt, nsrc * nper / dt, st["ms_fim_coarse"], nsrc * nper / (st["ms_fim_coarse"] / 1e3), nsrc * nper * (8.0 * e.nnx * e.nnz + 129 * 129 * 8) / (st["ms_fim_coarse"] / 1e3) / 1e9,
           st["ms_fim_refined"], st["ms_stages"], st["rounds_max"], st["evals_total"] / (nsrc * nper) / (e.nnx * e.nnz), st["chunk"], st["field_slots"], st["footprint_mb"] / 1e3, st["bundles"], st["bundle_size"], st["bundle_slots"]), flush=True)
e.close()
if nsample > 0:
    t = t.reshape(nsrc * nper, nrec)
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    veln = [L.o_gridder(g, pv[p]) for p in range(nper)]
    pick = np.linspace(0, nsrc * nper - 1, nsample).astype(int)

    def one(k):
        p = int(u["map_index"][k])
        o = L.o_solve(g, pv[p], veln[p], u["scx"][k], u["scz"][k])
        return np.array([L.o_srtimes(g, veln[p], o["T"], u["scx"][k], u["scz"][k], u["rcx"][k * nrec + r], u["rcz"][k * nrec + r]) for r in range(nrec)], np.float32)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(nsample, os.cpu_count() or 1)) as ex:
        ref = np.stack(list(ex.map(one, pick)))
    d = np.abs(t[pick].astype(np.float64) - ref.astype(np.float64))
    print("oracle (Fast Marching, %d host threads): %d units in %.1f s | %d receiver times: max |dt| %.3g s, beyond 1e-4 s: %d, not bit-identical: %d; largest time %.1f s" %
          (min(nsample, os.cpu_count() or 1), nsample, time.perf_counter() - t0, d.size, d.max(), int((d > 1e-4).sum()), int((t[pick].view(np.uint32) != ref.view(np.uint32)).sum()), ref.max()), flush=True)
