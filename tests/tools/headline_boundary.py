#!/usr/bin/env python3
"""The whole CalSurfG boundary at the headline size (no oracle: timing and sanity only):
nx = ny = 131 (1025^2 grid), nz = 9, 16 Rayleigh phase periods, 1000 sources per period, R receivers each."""
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


def main():
    nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = E.load_library()
    t0 = time.perf_counter()
    c = synth.boundary_case(nx=131, ny=131, nz=9, kRc=16, kRg=0, kLc=0, kLg=0, nsrc=nsrc, nrcf=nrec, dvd=0.01, ragged=False, stations=bool(int(os.environ.get('DSA_STATIONS', '1'))))
    c["tRc"] = np.linspace(2.0, 17.0, 16)
    print("case built in %.1f s: ndata %d, nparpi %d" % (time.perf_counter() - t0, c["ndata"], c["nparpi"]), flush=True)
    nd, npar = c["ndata"], c["nparpi"]
    cap = int(nd * 2200)
    i32 = lambda v: C.byref(C.c_int(int(v)))
    f32 = lambda v: C.byref(C.c_float(float(v)))
    iw = np.zeros(cap + 1, np.int32); rw = np.zeros(cap, np.float32); col = np.zeros(cap, np.int32); dsurf = np.zeros(nd, np.float32)
    os.environ["DSA_MAXNAR"] = str(cap)
    nar = C.c_int(0)
    for k in range(2):
        t0 = time.perf_counter()
        rc = lib.dsa_calsurfg(i32(c["nx"]), i32(c["ny"]), i32(c["nz"]), i32(npar), L.ptr(c["vels"]), L.ptr(iw), L.ptr(rw), L.ptr(col), L.ptr(dsurf),
                              f32(c["goxd"]), f32(c["gozd"]), f32(c["dvxd"]), f32(c["dvzd"]), i32(c["kRc"]), i32(0), i32(0), i32(0),
                              L.ptr(c["tRc"]), L.ptr(c["tRg"]), L.ptr(c["tLc"]), L.ptr(c["tLg"]), L.ptr(c["wavetype"]), L.ptr(c["igrt"]), L.ptr(c["periods"]),
                              L.ptr(c["depz"]), f32(c["minthk"]), L.ptr(c["scxf"]), L.ptr(c["sczf"]), L.ptr(c["rcxf"]), L.ptr(c["rczf"]), L.ptr(c["nrc1"]),
                              L.ptr(c["nsrcsurf1"]), i32(c["kmax"]), i32(c["nsrcsurf"]), i32(c["nrcf"]), C.byref(nar))
        dt = time.perf_counter() - t0
        print("dsa_calsurfg pass %d: rc %d (%s), %.2f s wall; %d solves, %d rays, nar %d (%.0f per row, %.2f GB of COO)" %
              (k, rc, lib.dsa_dropin_error().decode(), dt, 16 * nsrc, nd, nar.value, nar.value / max(nd, 1), nar.value * 12 / 1e9), flush=True)
    print("dsurf range %.3f .. %.3f s, rw range %.3g .. %.3g" % (dsurf.min(), dsurf.max(), rw[:nar.value].min(), rw[:nar.value].max()))
    if "--spmv" in sys.argv:
        n = nar.value
        e = E.Engine(0)
        t0 = time.perf_counter()
        e.spmv_load(nd, npar, rw[:n], iw[1:n + 1], col[:n])
        t1 = time.perf_counter()
        x = np.linspace(-1, 1, npar).astype(np.float32); y = np.linspace(1, -1, nd).astype(np.float32)
        for mode in (1, 2, 1, 2):
            t2 = time.perf_counter(); out = e.spmv(mode, x, y); t3 = time.perf_counter()
            print("device aprod mode %d: %.1f ms (%.0f GB/s on 8 B per entry)" % (mode, 1e3 * (t3 - t2), n * 8 / (t3 - t2) / 1e9), flush=True)
        print("matrix load (upload + stable radix sorts on the device, %d entries): %.2f s" % (n, t1 - t0))
        iwf = np.concatenate([[n], iw[1:n + 1], col[:n]]).astype(np.int32)
        ib = lambda v: C.byref(C.c_int(int(v)))
        for mode in (1, 2):
            xx, yy = x.copy(), y.copy()
            t2 = time.perf_counter()
            L.oracle().dso_aprod(ib(mode), ib(nd), ib(npar), L.ptr(xx), L.ptr(yy), ib(iwf.size), ib(n), L.ptr(iwf), L.ptr(rw))
            t3 = time.perf_counter()
            got = e.spmv(mode, x, y)
            want = yy if mode == 1 else xx
            print("CPU aprod (C restatement, one core) mode %d: %.1f ms; device result identical: %s" % (mode, 1e3 * (t3 - t2), bool((got.view(np.uint32) == want.view(np.uint32)).all())), flush=True)
        e.close()
    if "--lsmr" in sys.argv:
        # one outer iteration's inversion step at the headline size: system of main.f90:361-466 on the host, LSMR on the device
        n0 = nar.value
        maxvp = npar
        cap2 = n0 + 7 * maxvp
        rw2 = np.zeros(cap2, np.float32); rw2[:n0] = rw[:n0]
        col2 = np.zeros(cap2, np.int32); col2[:n0] = col[:n0]
        iw2 = np.zeros(2 * cap2 + 1, np.int32); iw2[1:n0 + 1] = iw[1:n0 + 1]
        r = synth.LCG(9)
        obst = (dsurf * (1.0 + 0.02 * (r.uniform(nd) - 0.5))).astype(np.float32)
        cbst = np.zeros(nd + maxvp, np.float32); dw = np.zeros(nd, np.float32); norm = np.zeros(maxvp, np.float32); dws = np.zeros(2, np.float32)
        m, n2 = C.c_int(0), C.c_longlong(0)
        lib.dsa_iteration_system.argtypes = [C.c_int] * 4 + [C.c_longlong] * 2 + [C.c_void_p] * 5 + [C.c_float] * 2 + [C.c_void_p] * 6
        t0 = time.perf_counter()
        rc = lib.dsa_iteration_system(c["nx"], c["ny"], c["nz"], nd, n0, cap2, L.ptr(rw2), L.ptr(iw2), L.ptr(col2), L.ptr(obst), L.ptr(dsurf), 3.0, 4.0,
                                      L.ptr(cbst), L.ptr(dw), L.ptr(norm), C.byref(m), C.byref(n2), L.ptr(dws))
        print("dsa_iteration_system (host): rc %d, %.2f s; m %d n %d nar %d" % (rc, time.perf_counter() - t0, m.value, maxvp, n2.value), flush=True)
        n = n2.value
        e = E.Engine(0)
        t0 = time.perf_counter()
        e.spmv_load(m.value, maxvp, rw2[:n], iw2[1:n + 1], iw2[n + 1:2 * n + 1])
        print("matrix load: %.2f s" % (time.perf_counter() - t0), flush=True)
        for dvec in (0, 1):
            e.set_option("lsmr_device_vectors", dvec)
            name = "vectors on the device" if dvec else "products on the device, ordered sums on the host"
            e.lsmr(cbst, 1.0, itnlim=2)
            t0 = time.perf_counter(); got = e.lsmr(cbst, 1.0, itnlim=30); dt = time.perf_counter() - t0
            print("LSMR (%s): %d iterations in %.3f s (%.2f ms per iteration)" % (name, got["itn"], dt, 1e3 * dt / max(got["itn"], 1)), flush=True)
            t0 = time.perf_counter(); full = e.lsmr(cbst, 1.0); dt = time.perf_counter() - t0
            print("LSMR (%s) to convergence: %d iterations, istop %d, %.3f s; |dv| max %.4f" % (name, full["itn"], full["istop"], dt, np.abs(full["x"]).max()), flush=True)
        e.set_option("lsmr_device_vectors", 0)
        import inversion as inv
        S = dict(m=m.value, n=maxvp, iw=iw2[:2 * n + 1], rw=rw2[:n], b=cbst)
        t0 = time.perf_counter(); want = inv.call_lsmr(L.oracle().dso_lsmr, S, 1.0, itnlim=3); dt = time.perf_counter() - t0
        got = e.lsmr(cbst, 1.0, itnlim=3)
        print("CPU LSMR (C restatement, one core): 3 iterations in %.2f s (%.0f ms per iteration); device result identical: %s" %
              (dt, 1e3 * dt / 3, inv.same(got, want) == []), flush=True)
        e.close()


    if "--device-rows" in sys.argv:
        # the same outer iteration with the matrix resident on the device from CalSurfG to LSMR
        from dsurftomo_amd import invert
        invert.bind(lib)
        eng = lib.dsa_dropin_engine()
        dsurf2 = np.zeros(nd, np.float32)
        nar_d = C.c_int(0)
        lib.dsa_dropin_set_capacity(0)
        for k in range(2):
            t0 = time.perf_counter()
            rc = lib.dsa_calsurfg(i32(c["nx"]), i32(c["ny"]), i32(c["nz"]), i32(npar), L.ptr(c["vels"]), None, None, None, L.ptr(dsurf2),
                                  f32(c["goxd"]), f32(c["gozd"]), f32(c["dvxd"]), f32(c["dvzd"]), i32(c["kRc"]), i32(0), i32(0), i32(0),
                                  L.ptr(c["tRc"]), L.ptr(c["tRg"]), L.ptr(c["tLc"]), L.ptr(c["tLg"]), L.ptr(c["wavetype"]), L.ptr(c["igrt"]), L.ptr(c["periods"]),
                                  L.ptr(c["depz"]), f32(c["minthk"]), L.ptr(c["scxf"]), L.ptr(c["sczf"]), L.ptr(c["rcxf"]), L.ptr(c["rczf"]), L.ptr(c["nrc1"]),
                                  L.ptr(c["nsrcsurf1"]), i32(c["kmax"]), i32(c["nsrcsurf"]), i32(c["nrcf"]), C.byref(nar_d))
            print("dsa_calsurfg, rows left on the device, pass %d: rc %d, %.2f s wall, nar %d; dsurf identical to the host-array call: %s" %
                  (k, rc, time.perf_counter() - t0, nar_d.value, bool((dsurf2.view(np.uint32) == dsurf.view(np.uint32)).all())), flush=True)
        r = synth.LCG(9)
        obst = (dsurf * (1.0 + 0.02 * (r.uniform(nd) - 0.5))).astype(np.float32)
        maxvp = npar
        cbst_d = np.zeros(nd + maxvp, np.float32); dw_d = np.zeros(nd, np.float32); norm_d = np.zeros(maxvp, np.float32); dws_d = np.zeros(2, np.float32)
        m, n2 = C.c_int(0), C.c_longlong(0)
        t0 = time.perf_counter()
        rc = lib.dsa_iteration_system_device(eng, c["nx"], c["ny"], c["nz"], nd, L.ptr(obst), L.ptr(dsurf2), 3.0, 4.0, L.ptr(cbst_d), L.ptr(dw_d), L.ptr(norm_d),
                                             C.byref(m), C.byref(n2), L.ptr(dws_d))
        print("dsa_iteration_system_device: rc %d, %.2f s (weights, regularisation rows, both orderings, DWS); m %d nar %d; DWS %g %g" %
              (rc, time.perf_counter() - t0, m.value, n2.value, dws_d[0], dws_d[1]), flush=True)
        if "--lsmr" in sys.argv:
            print("  same as the host system: cbst %s datweight %s norm %s dws %s" % (np.array_equal(cbst_d.view(np.uint32), cbst.view(np.uint32)),
                  np.array_equal(dw_d, dw), np.array_equal(norm_d.view(np.uint32), norm.view(np.uint32)), bool((dws_d == dws).all())), flush=True)
        dv = np.zeros(maxvp, np.float32)
        ii = [C.c_int(0), C.c_int(0)]; ff = [C.c_float(0) for _ in range(5)]
        t0 = time.perf_counter()
        rc = lib.dsa_lsmr(eng, L.ptr(cbst_d), C.c_float(1.0), C.c_float(1e-6), C.c_float(1e-6), C.c_float(100.0), 400, 10, L.ptr(dv),
                          C.byref(ii[0]), C.byref(ii[1]), *[C.byref(v) for v in ff])
        dt = time.perf_counter() - t0
        print("dsa_lsmr on the resident matrix: rc %d, %d iterations, istop %d, %.3f s (%.2f ms per iteration); |dv| max %.4f" %
              (rc, ii[1].value, ii[0].value, dt, 1e3 * dt / max(ii[1].value, 1), np.abs(dv).max()), flush=True)
        if "--lsmr" in sys.argv:
            print("  solution identical to the host-array path: %s" % bool((dv.view(np.uint32) == full["x"].view(np.uint32)).all()), flush=True)


if __name__ == "__main__":
    main()
