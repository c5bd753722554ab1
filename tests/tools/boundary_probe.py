#!/usr/bin/env python3
"""Whole-boundary probe on the GPU: dsa_calsurfg / dsa_synthetic / dispersion stage against the oracle.

    python tests/tools/boundary_probe.py [case ...]      (cases: default deep groups big)
Prints how many outputs are bit-identical and the largest differences.  Test infrastructure use of
the oracle (tools are not part of the product path).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L      # noqa: E402
import synth           # noqa: E402
from dsurftomo_amd import engine as E   # noqa: E402

CASES = {
    "default": dict(),
    "deep": dict(deep=True, nz=7, nx=10, ny=13, seed=5),
    "groups": dict(kRc=0, kRg=2, kLc=0, kLg=1),
    "big": dict(nx=20, ny=18, nz=6, nsrc=8, nrcf=7, kRc=3, kRg=1, kLc=1, kLg=1),
}


def dense(r, ndata, npar):
    G = np.zeros((ndata, npar), np.float32)
    G[r["iw"] - 1, r["col"] - 1] = r["rw"]
    return G


def main():
    lib = E.load_library()
    names = sys.argv[1:] or ["default", "deep", "groups"]
    for name in names:
        c = synth.boundary_case(**CASES[name])
        t0 = time.time()
        o = L.call_boundary(L.oracle().dso_calsurfg, c)
        t1 = time.time()
        try:
            d = L.call_boundary(lib.dsa_calsurfg, c)
        except Exception as ex:
            print(name, "FAILED", ex, lib.dsa_dropin_error())
            continue
        t2 = time.time()
        err = lib.dsa_dropin_error().decode()
        print("== %s: ndata %d nparpi %d | oracle %.2fs device %.2fs %s" % (name, c["ndata"], c["nparpi"], t1 - t0, t2 - t1, err))
        dt = np.abs(o["dsurf"] - d["dsurf"])
        print("   dsurf: bit-identical %d/%d, max |d| %.3g s" % (int((o["dsurf"].view(np.uint32) == d["dsurf"].view(np.uint32)).sum()), dt.size, dt.max()))
        print("   nar: oracle %d device %d" % (o["nar"], d["nar"]))
        Go, Gd = dense(o, c["ndata"], c["nparpi"]), dense(d, c["ndata"], c["nparpi"])
        dG = np.abs(Go - Gd)
        pat = (Go != 0) != (Gd != 0)
        print("   G: entries identical %.4f%%, pattern differences %d, max |dG| %.3g (max |G| %.3g), rel fro %.3g" %
              (100.0 * (Go.view(np.uint32) == Gd.view(np.uint32)).mean(), int(pat.sum()), dG.max(), np.abs(Go).max(),
               np.linalg.norm(Go - Gd) / np.linalg.norm(Go)))
        if pat.sum():
            vals = np.where(Go != 0, Go, Gd)[pat]
            print("   values at pattern differences: max |v| %.3g" % np.abs(vals).max())
        so = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True)
        sd = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
        print("   synthetic: bit-identical %d/%d, max |d| %.3g s" % (int((so.view(np.uint32) == sd.view(np.uint32)).sum()), so.size, np.abs(so - sd).max()))
    # dispersion stage alone
    c = synth.boundary_case(nx=12, ny=10, nz=7)
    vel = np.ascontiguousarray(c["vels"].T)
    t = np.array([1.0, 2.0, 4.0, 7.0, 11.0, 16.0, 22.0])
    e = E.Engine(0)
    for iwave in (2, 1):
        for igr in (0, 1):
            ref = L.depthkernel("oracle", vel, c["depz"], float(c["minthk"]), iwave, igr, t)
            e.dispersion_begin(vel, c["depz"], float(c["minthk"]), len(t), len(t))
            t0 = time.time()
            e.dispersion_run(iwave, igr, t, True, 0, 0)
            dt = time.time() - t0
            dev = e.dispersion_fetch(0, len(t), True, 0)
            same = [float((np.ascontiguousarray(a).view(np.uint64) == np.ascontiguousarray(b).view(np.uint64)).mean()) for a, b in zip(dev, ref)]
            mx = [float(np.abs(a - b).max()) for a, b in zip(dev, ref)]
            print("dispersion iwave %d igr %d: %.1f ms; identical fraction pv %.5f sen %s; max |d| pv %.3g sen %s; max |sen| %.3g" %
                  (iwave, igr, 1000 * dt, same[0], ["%.5f" % s for s in same[1:]], mx[0], ["%.3g" % m for m in mx[1:]], max(np.abs(r).max() for r in ref[1:])))
    print("stats", {k: v for k, v in e.stats().items() if k in ("ms_dispersion", "curves")})
    e.close()


if __name__ == "__main__":
    main()
