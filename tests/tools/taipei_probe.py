#!/usr/bin/env python3
"""Taipei example (BASELINE.json configs[0]) end to end: one CalSurfG call on the device (drop-in
entry) timed next to the reference's own Fortran (oracle/_ref, if present) on this box's cores."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L      # noqa: E402
from dsurftomo_amd import io as taipei          # noqa: E402
from dsurftomo_amd import engine as E   # noqa: E402


def main():
    lib = E.load_library()
    c = taipei.load()
    print("Taipei: nx %d ny %d nz %d, %d periods, %d data, nparpi %d" % (c["nx"], c["ny"], c["nz"], c["kmax"], c["ndata"], c["nparpi"]))
    for k in range(3):
        t0 = time.perf_counter()
        d = L.call_boundary(lib.dsa_calsurfg, c)
        dt = time.perf_counter() - t0
        print("device call %d: %.3f s wall, nar %d" % (k, dt, d["nar"]))
        st = np.zeros(64)
        lib.dsa_get_stats.argtypes = [C.c_void_p, C.c_void_p]
    # stage times of the last call (the drop-in's process-wide engine is not exposed; time the stages through an own engine)
    e = E.Engine(0)
    if os.environ.get("DSA_PROBE_TIES"):
        e.set_option("exact_ties", int(os.environ["DSA_PROBE_TIES"]))
    vel = np.ascontiguousarray(c["vels"].T)
    for k in range(2):
        t0 = time.perf_counter()
        e.dispersion_begin(vel, c["depz"], float(c["minthk"]), c["kmax"], c["kmax"])
        e.dispersion_run(2, 0, c["tRc"], True, 0, 0)
        t1 = time.perf_counter()
        e.maps_from_dispersion(c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], 8)
        e.kernels_from_dispersion()
        t2 = time.perf_counter()
        maps, sx, sz, nrec, rx, rz, slot = [], [], [], [], [], [], []
        for kk in range(c["kmax"]):
            for s in range(c["nsrcsurf1"][kk]):
                maps.append(c["periods"][s, kk] - 1); sx.append(c["scxf"][s, kk]); sz.append(c["sczf"][s, kk]); slot.append(kk)
                nrec.append(c["nrc1"][s, kk]); rx += list(c["rcxf"][:nrec[-1], s, kk]); rz += list(c["rczf"][:nrec[-1], s, kk])
        e.plan(maps, sx, sz, nrec, rx, rz, sen_slot=slot)
        t3 = time.perf_counter()
        out = e.solve_rows(c["ndata"] * c["nparpi"])
        t4 = time.perf_counter()
        st = e.stats()
        print("stages (own engine) pass %d: dispersion %.1f ms (kernel %.1f, %d curves), maps+kernels %.1f ms, plan %.1f ms (%d units), solve_rows %.1f ms "
              "[fim coarse %.1f refined %.1f stages %.1f rays %.1f rows %.1f]" % (k, 1e3 * (t1 - t0), st["ms_dispersion"], st["curves"], 1e3 * (t2 - t1), 1e3 * (t3 - t2), len(maps),
              1e3 * (t4 - t3), st["ms_fim_coarse"], st["ms_fim_refined"], st["ms_stages"], st["ms_rays"], st["ms_rows"]))
        print("    tie handling (exact_ties = %s): %d units flagged, %d marched in %.1f ms, largest influence %.3g s" % (os.environ.get("DSA_PROBE_TIES", "default"), st["tie_units"], st["exact_units"], st["ms_exact"], st.get("tie_influence_max", 0.0)))
    e.close()
    ref = L.ref()
    if ref is not None and "--no-ref" not in sys.argv:
        for threads in (os.cpu_count(), 1):
            os.environ["OMP_NUM_THREADS"] = str(threads)
            try:
                C.CDLL("libomp.so").omp_set_num_threads(threads)
            except OSError:
                pass
            t0 = time.perf_counter()
            a = L.call_boundary(ref.calsurfg_, c)
            dt = time.perf_counter() - t0
            print("reference CalSurfG, %d thread(s): %.2f s, nar %d" % (threads, dt, a["nar"]))
        print("max |dsurf diff| %.3g s; nar equal %s" % (np.abs(a["dsurf"] - d["dsurf"]).max(), a["nar"] == d["nar"]))
        if a["nar"] == d["nar"]:
            print("rw identical %d / %d, col identical %s, iw identical %s" % (int((a["rw"].view(np.uint32) == d["rw"].view(np.uint32)).sum()), a["nar"],
                  bool((a["col"] == d["col"]).all()), bool((a["iw"] == d["iw"]).all())))


if __name__ == "__main__":
    main()
