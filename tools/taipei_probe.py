#!/usr/bin/env python3
"""Taipei example (BASELINE.json configs[0]) end to end: one CalSurfG call on the device (drop-in
entry) timed next to the reference's own Fortran (oracle/_ref, if present) on this box's cores."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L      # noqa: E402
import taipei          # noqa: E402
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
