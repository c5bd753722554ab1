"""The reference's inversion loop (main.f90:346-590) driven through this library, without the Fortran host program.

    python -m dsurftomo_amd.invert <directory with DSurfTomo.in, the data file and MOD> [--maxiter N] [--out DIR]

Per outer iteration: CalSurfG on the device (dsa_calsurfg: dispersion, depth kernels, eikonal solves, rays, Frechet rows),
the glue of main.f90:361-466 (residuals, percentile weights, DWS, regularisation rows), LSMR on the device (bit-identical
to the reference's LSMR), the model update of main.f90:520-535.  By default the matrix never leaves the device
(dsa_calsurfg with null rw / iw / col -> dsa_iteration_system_device -> dsa_lsmr); --host-rows hands it through host
arrays the way the reference does (dsa_calsurfg -> dsa_iteration_system -> dsa_lsmr_dropin), with the same numbers.  Then
(dsa_model_update), and the reference's output files: residualFirst.dat / residualLast.dat (main.f90:397-411),
<input>Measure.dat.iterNNN (main.f90:537-546) and <input>Measure.dat (:575-584), in the reference's formats.
Synthetic tests (ifsyn = 1, main.f90:326-343) forward-model MOD.true; the noise there comes from this module's own
generator, not from the reference's gaussian().  There is no CPU path: without a usable GPU this fails with the engine's
error text.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

from . import io
from .engine import load_library


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f10(v):
    return "%10.5f" % v


def write_model(path, c, vsf):
    """'(5f10.5)' lines: longitude, latitude, depth, Vs for the interior vertices, k / j / i order (main.f90:539-545)"""
    nx, ny, nz = c["nx"], c["ny"], c["nz"]
    f = np.float32
    with open(path, "w") as fh:
        for k in range(nz - 1):
            for j in range(ny - 2):
                for i in range(nx - 2):
                    lon = f(c["gozd"] + f(f(j) * c["dvzd"]))
                    lat = f(c["goxd"] - f(f(i) * c["dvxd"]))
                    fh.write(_f10(lon) + _f10(lat) + _f10(c["depz"][k]) + _f10(vsf[i + 1, j + 1, k]) + "\n")


def write_residuals(path, c, dsyn, obst, datweight):
    """list-directed rows: dist, dsyn, obst, dsyn*w, obst*w, w (main.f90:397-403)"""
    np.savetxt(path, np.column_stack([c["dist"], dsyn, obst, dsyn * datweight, obst * datweight, datweight]), fmt="%16.8f")


def iteration_device(lib, c, vsf, obst, log):
    """One pass of main.f90:349-535 with the matrix resident on the device from CalSurfG to LSMR: dsa_calsurfg leaves the
    rows there (null rw / iw / col), dsa_iteration_system_device applies weights / appends the regularisation rows / builds
    both orderings in place, dsa_lsmr solves.  Same numbers as iteration() (tests/test_gpu_lsmr.py compares every bit)."""
    f = np.float32
    nx, ny, nz, dall = c["nx"], c["ny"], c["nz"], c["ndata"]
    maxvp = c["nparpi"]
    dsyn = np.zeros(dall, f)
    nar = C.c_int(0)
    cc = dict(c); cc["vels"] = vsf
    head, tail = io._args(cc)
    lib.dsa_dropin_set_capacity(0)
    t0 = time.perf_counter()
    if lib.dsa_calsurfg(*head, None, None, None, _p(dsyn), *tail, C.byref(nar)) != 0:
        raise RuntimeError("dsa_calsurfg: %s" % lib.dsa_dropin_error().decode())
    t_fwd = time.perf_counter() - t0
    eng = lib.dsa_dropin_engine()
    cbst = np.zeros(dall + maxvp, f); datweight = np.zeros(dall, f); norm = np.zeros(maxvp, f); dws = np.zeros(2, f)
    m, nar2 = C.c_int(0), C.c_longlong(0)
    t0 = time.perf_counter()
    rc = lib.dsa_iteration_system_device(eng, nx, ny, nz, dall, _p(obst), _p(dsyn), c["threshold0"], c["weight0"], _p(cbst), _p(datweight), _p(norm),
                                         C.byref(m), C.byref(nar2), _p(dws))
    if rc != 0:
        raise RuntimeError("dsa_iteration_system_device failed (%d): %s" % (rc, lib.dsa_error_string(eng).decode()))
    t_glue = time.perf_counter() - t0
    log("Maximum and Average DWS values: %g %g" % (dws[0], dws[1]))
    dv = np.zeros(maxvp, f)
    ii = [C.c_int(0), C.c_int(0)]
    ff = [C.c_float(0) for _ in range(5)]
    t0 = time.perf_counter()
    rc = lib.dsa_lsmr(eng, _p(cbst), C.c_float(c["damp"]), C.c_float(1e-6), C.c_float(1e-6), C.c_float(100.0), 400, 10, _p(dv),
                      C.byref(ii[0]), C.byref(ii[1]), *[C.byref(v) for v in ff])
    if rc != 0:
        raise RuntimeError("dsa_lsmr: %s" % lib.dsa_error_string(eng).decode())
    t_lsmr = time.perf_counter() - t0
    r = cbst[:dall]
    mean = f(r.sum(dtype=f) / f(dall))
    std = f(np.sqrt(f((r * r).sum(dtype=f) / f(dall)) - mean * mean))
    rms = f(np.sqrt((r.astype(np.float64) ** 2).sum()) / np.sqrt(dall))
    dv_raw = (f(dv.min()), f(dv.max()))
    lib.dsa_model_update(nx, ny, nz, _p(dv), _p(vsf), c["minvel"], c["maxvel"])
    return dict(dsyn=dsyn, datweight=datweight, mean_ms=1e3 * float(mean), std_ms=1e3 * float(std), rms=float(rms), dv_min=float(dv_raw[0]),
                dv_max=float(dv_raw[1]), itn=ii[1].value, istop=ii[0].value, nar=nar2.value, m=m.value, dws=(float(dws[0]), float(dws[1])),
                seconds=dict(forward=t_fwd, glue=t_glue, lsmr=t_lsmr), dv=dv, norm=norm, cbst=cbst)


def iteration(lib, c, vsf, obst, log):
    """One pass of main.f90:349-535 on the model vsf (updated in place), the matrix going through host arrays like in the
    reference (dsa_calsurfg -> dsa_iteration_system -> dsa_lsmr_dropin).  Returns the statistics of the pass."""
    f = np.float32
    nx, ny, nz, dall = c["nx"], c["ny"], c["nz"], c["ndata"]
    maxvp = c["nparpi"]
    maxnar = int(f(c["spfra"]) * dall * nx * ny * nz)                                  # main.f90:287
    rw = np.zeros(maxnar, f); col = np.zeros(maxnar, np.int32); iw = np.zeros(2 * maxnar + 1, np.int32)
    dsyn = np.zeros(dall, f)
    nar = C.c_int(0)
    cc = dict(c); cc["vels"] = vsf
    head, tail = io._args(cc)
    lib.dsa_dropin_set_capacity(maxnar)
    t0 = time.perf_counter()
    if lib.dsa_calsurfg(*head, _p(iw), _p(rw), _p(col), _p(dsyn), *tail, C.byref(nar)) != 0:
        raise RuntimeError("dsa_calsurfg: %s" % lib.dsa_dropin_error().decode())
    t_fwd = time.perf_counter() - t0
    cbst = np.zeros(dall + maxvp, f); datweight = np.zeros(dall, f); norm = np.zeros(maxvp, f); dws = np.zeros(2, f)
    m, nar2 = C.c_int(0), C.c_longlong(0)
    t0 = time.perf_counter()
    rc = lib.dsa_iteration_system(nx, ny, nz, dall, nar.value, maxnar, _p(rw), _p(iw), _p(col), _p(obst), _p(dsyn), c["threshold0"], c["weight0"],
                                  _p(cbst), _p(datweight), _p(norm), C.byref(m), C.byref(nar2), _p(dws))
    if rc != 0:
        raise RuntimeError("increase sparsity fraction(spfra)" if rc == -6 else "dsa_iteration_system failed (%d)" % rc)
    t_glue = time.perf_counter() - t0
    log("Maximum and Average DWS values: %g %g" % (dws[0], dws[1]))
    dv = np.zeros(maxvp, f)
    ii = [C.c_int(0), C.c_int(0)]
    ff = [C.c_float(0) for _ in range(5)]
    i32 = lambda v: C.byref(C.c_int(int(v)))
    f32 = lambda v: C.byref(C.c_float(float(v)))
    n = nar2.value
    t0 = time.perf_counter()
    rc = lib.dsa_lsmr_dropin(i32(m.value), i32(maxvp), i32(2 * n + 1), i32(n), _p(iw), _p(rw), _p(cbst), f32(c["damp"]), f32(1e-6), f32(1e-6),
                             f32(100.0), i32(400), i32(10), i32(0), _p(dv), C.byref(ii[0]), C.byref(ii[1]), *[C.byref(v) for v in ff])
    if rc != 0:
        raise RuntimeError("dsa_lsmr_dropin: %s" % lib.dsa_dropin_error().decode())
    t_lsmr = time.perf_counter() - t0
    r = cbst[:dall]
    mean = f(r.sum(dtype=f) / f(dall))
    std = f(np.sqrt(f((r * r).sum(dtype=f) / f(dall)) - mean * mean))
    rms = f(np.sqrt((r.astype(np.float64) ** 2).sum()) / np.sqrt(dall))
    dv_raw = (f(dv.min()), f(dv.max()))
    lib.dsa_model_update(nx, ny, nz, _p(dv), _p(vsf), c["minvel"], c["maxvel"])
    return dict(dsyn=dsyn, datweight=datweight, mean_ms=1e3 * float(mean), std_ms=1e3 * float(std), rms=float(rms), dv_min=float(dv_raw[0]),
                dv_max=float(dv_raw[1]), itn=ii[1].value, istop=ii[0].value, nar=n, m=m.value, dws=(float(dws[0]), float(dws[1])),
                seconds=dict(forward=t_fwd, glue=t_glue, lsmr=t_lsmr), dv=dv, norm=norm, cbst=cbst)


def bind(lib):
    lib.dsa_iteration_system.argtypes = [C.c_int] * 4 + [C.c_longlong] * 2 + [C.c_void_p] * 5 + [C.c_float] * 2 + [C.c_void_p] * 6
    lib.dsa_iteration_system_device.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_float] * 2 + [C.c_void_p] * 6
    lib.dsa_model_update.argtypes = [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_float] * 2
    lib.dsa_dropin_engine.restype = C.c_void_p
    lib.dsa_dropin_engine.argtypes = []
    lib.dsa_lsmr.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_int] * 2 + [C.c_void_p] * 8
    lib.dsa_error_string.restype = C.c_char_p
    lib.dsa_error_string.argtypes = [C.c_void_p]
    return lib


def run(directory, maxiter=None, out_dir=".", log=print, seed=1, host_rows=False):
    lib = bind(load_library())
    c = io.load(directory)
    maxiter = c["maxiter"] if maxiter is None else maxiter
    vsf = np.asfortranarray(c["vels"].copy())
    obst = np.ascontiguousarray(c["obst"])
    vsftrue = None
    if c["ifsyn"] == 1:                                                                 # main.f90:326-343
        vsftrue = io.load(directory, "MOD.true")["vels"]
        ct = dict(c); ct["vels"] = vsftrue
        obst = io.call_synthetic(ct, 0.0)
        g = np.random.default_rng(seed).standard_normal(obst.size).astype(np.float32)
        obst = (obst * (np.float32(1.0) + c["noiselevel"] * g)).astype(np.float32)
    name = os.path.join(out_dir, "DSurfTomo.in")
    history = []
    for it in range(1, maxiter + 1):
        st = (iteration if host_rows else iteration_device)(lib, c, vsf, obst, log)
        log("%2dth iteration..." % it)
        log(" mean,std_devs and rms of residual after weighting: %8.1fms %8.2fms %8.3f" % (st["mean_ms"], st["std_ms"], st["rms"]))
        log(" min and max velocity variation %7.4f%7.4f" % (st["dv_min"], st["dv_max"]))
        log("   (forward %.3f s, system %.3f s, LSMR %.3f s: %d iterations, istop %d, %d x %d, %d entries)" %
            (st["seconds"]["forward"], st["seconds"]["glue"], st["seconds"]["lsmr"], st["itn"], st["istop"], st["m"], c["nparpi"], st["nar"]))
        if it == 1:
            write_residuals(os.path.join(out_dir, "residualFirst.dat"), c, st["dsyn"], obst, st["datweight"])
        if it == maxiter:
            write_residuals(os.path.join(out_dir, "residualLast.dat"), c, st["dsyn"], obst, st["datweight"])
        write_model(name + "Measure.dat.iter%03d" % it, c, vsf)
        history.append({k: v for k, v in st.items() if k not in ("dsyn", "datweight", "dv", "norm", "cbst")})
    if vsftrue is not None:
        write_model(os.path.join(out_dir, "Vs_model.real"), c, vsftrue)
        write_model(name + "Syn.dat", c, vsf)
    else:
        write_model(name + "Measure.dat", c, vsf)
    log("Program finishes successfully")
    return vsf, history


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("directory")
    ap.add_argument("--maxiter", type=int, default=None)
    ap.add_argument("--out", default=".")
    ap.add_argument("--host-rows", action="store_true", help="hand the matrix through host arrays like the reference (default: it stays on the device)")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    run(args.directory, args.maxiter, args.out, host_rows=args.host_rows)
    return 0


if __name__ == "__main__":
    sys.exit(main())
