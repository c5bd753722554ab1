#!/usr/bin/env python3
"""Stage times of one forward call at the headline size through an own engine (no oracle involved):
nx = ny = 131, nz = 9, 16 Rayleigh phase periods, <nsrc> sources per period, <nrec> receivers each."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth           # noqa: E402
from dsurftomo_amd import engine as E   # noqa: E402

nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
c = synth.boundary_case(nx=131, ny=131, nz=9, kRc=16, kRg=0, kLc=0, kLg=0, nsrc=nsrc, nrcf=nrec, dvd=0.01, ragged=False, stations=bool(int(os.environ.get('DSA_STATIONS', '1'))))
c["tRc"] = np.linspace(2.0, 17.0, 16)
e = E.Engine(0)
if os.environ.get("DSA_BUNDLE"): e.set_option("bundle", int(os.environ["DSA_BUNDLE"]))
if os.environ.get("DSA_RAY_BUDGET_GB"): e.set_option("ray_budget", float(os.environ["DSA_RAY_BUDGET_GB"]) * 1e9)
vel = np.ascontiguousarray(c["vels"].T)
maps, sx, sz, nr, rx, rz, slot = [], [], [], [], [], [], []
for kk in range(c["kmax"]):
    for s in range(c["nsrcsurf1"][kk]):
        maps.append(c["periods"][s, kk] - 1); sx.append(c["scxf"][s, kk]); sz.append(c["sczf"][s, kk]); slot.append(kk)
        nr.append(c["nrc1"][s, kk]); rx += list(c["rcxf"][:nr[-1], s, kk]); rz += list(c["rczf"][:nr[-1], s, kk])
for k in range(2):
    t0 = time.perf_counter()
    e.dispersion_begin(vel, c["depz"], float(c["minthk"]), c["kmax"], c["kmax"])
    e.dispersion_run(2, 0, c["tRc"], True, 0, 0)
    t1 = time.perf_counter()
    e.maps_from_dispersion(c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], 8)
    e.kernels_from_dispersion()
    t2 = time.perf_counter()
    e.plan(maps, sx, sz, nr, rx, rz, sen_slot=slot)
    t3 = time.perf_counter()
    out = e.solve_rows(int(c["ndata"] * 2200))
    t4 = time.perf_counter()
    st = e.stats()
    print("pass %d: dispersion %.0f ms, maps+kernels %.0f ms, plan %.0f ms (%d units), solve_rows %.0f ms wall [solve kernels: coarse %.0f refined %.0f "
          "stages %.0f | rays %.0f rows (incl. copy out) %.0f], nar %d; bundles %d of %d members (%d units), distinct sources %d" %
          (k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), len(maps), 1e3 * (t4 - t3), st["ms_fim_coarse"], st["ms_fim_refined"], st["ms_stages"],
           st["ms_rays"], st["ms_rows"], out[1].size, st["bundles"], st["bundle_size"], st["bundled_units"], len(set(zip(np.asarray(sx, np.float32).tolist(), np.asarray(sz, np.float32).tolist())))), flush=True)
e.close()
