"""Readers for the reference's on-disk inputs: the parameter file (DSurfTomo.in), the measurement
file (`#`-headed source blocks) and the model file (MOD) -- reference main.f90:134-335.

`load(directory)` builds the argument set of a CalSurfG / synthetic call exactly as the reference's
host program does: fp32 colatitude / longitude in radians with pi = 3.1415926535898, period slots
Rc | Rg | Lc | Lg, sources counted per slot in file order, observed times `dist / velocity` with the
reference's `delsph` distance.  The returned dict is what `call_calsurfg` / `call_synthetic` (ctypes
bindings of the drop-in entries) take.  The reference's Taipei example lives under
tests/golden/taipei/ (BASELINE.json configs[0]).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "taipei")


def delsph(flat1, flon1, flat2, flon2):
    """great-circle distance (km) of the reference's delsph.f90 (haversine on colatitude / longitude in
    radians, R = 6371), in fp32 to the rounding of numpy's single-precision functions"""
    f = np.float32
    pi = f(3.1415926535898)
    dlat, dlon = f(flat2 - flat1), f(flon2 - flon1)
    lat1, lat2 = f(pi / f(2) - flat1), f(pi / f(2) - flat2)
    a = f(np.sin(dlat / f(2)) * np.sin(dlat / f(2)) + np.sin(dlon / f(2)) * np.sin(dlon / f(2)) * np.cos(lat1) * np.cos(lat2))
    return f(f(6371.0) * f(2) * np.arctan2(np.sqrt(a), np.sqrt(f(1) - a)))


def _vals(line):
    return line.split("c:")[0].split()


def load(directory=HERE, model="MOD"):
    f = np.float32
    with open(os.path.join(directory, "DSurfTomo.in")) as fh:
        lines = fh.read().splitlines()[3:]
    it = iter(lines)
    datafile = _vals(next(it))[0]
    nx, ny, nz = (int(v) for v in _vals(next(it))[:3])
    goxd, gozd = (f(v) for v in _vals(next(it))[:2])
    dvxd, dvzd = (f(v) for v in _vals(next(it))[:2])
    nsrc = int(_vals(next(it))[0])
    weight0, damp = (f(v) for v in _vals(next(it))[:2])
    minthk = f(_vals(next(it))[0])             # "sablayers"
    minvel, maxvel = (f(v) for v in _vals(next(it))[:2])
    maxiter = int(_vals(next(it))[0])
    spfra = float(_vals(next(it))[0])
    per = []
    for _ in range(4):
        k = int(_vals(next(it))[0])
        per.append(np.array([float(v) for v in next(it).split()[:k]], np.float64) if k > 0 else np.zeros(0))
    ifsyn = int(_vals(next(it))[0])
    noiselevel = f(_vals(next(it))[0])
    nxt = next(it, None)
    threshold0 = f(_vals(nxt)[0]) if nxt is not None and _vals(nxt) else f(0.0)      # main.f90:210
    kRc, kRg, kLc, kLg = (len(p) for p in per)
    kmax = kRc + kRg + kLc + kLg
    nrc = nsrc
    pi = f(3.1415926535898)
    scxf = np.zeros((nsrc, kmax), f, order="F"); sczf = np.zeros((nsrc, kmax), f, order="F")
    rcxf = np.zeros((nrc, nsrc, kmax), f, order="F"); rczf = np.zeros((nrc, nsrc, kmax), f, order="F")
    periods = np.zeros((nsrc, kmax), np.int32, order="F"); wavetype = np.zeros((nsrc, kmax), np.int32, order="F")
    igrt = np.zeros((nsrc, kmax), np.int32, order="F"); nrc1 = np.zeros((nsrc, kmax), np.int32, order="F")
    nsrc1 = np.zeros(kmax, np.int32)
    vel_obs, dist = [], []
    src_lat = src_lon = f(0)
    istep = istep1 = 0
    knum = 0
    knumo = 12345
    with open(os.path.join(directory, datafile)) as fh:
        for line in fh:
            if not line.strip():
                continue
            if line[0] == "#":
                t = line[1:].split()
                lat, lon, period, wavetp, veltp = f(t[0]), f(t[1]), int(t[2]), int(t[3]), int(t[4])
                if wavetp == 2 and veltp == 0: knum = period
                if wavetp == 2 and veltp == 1: knum = kRc + period
                if wavetp == 1 and veltp == 0: knum = kRg + kRc + period
                if wavetp == 1 and veltp == 1: knum = kLc + kRg + kRc + period
                if knum != knumo:
                    istep = 0
                istep += 1
                istep1 = 0
                src_lat = (f(90.0) - lat) * pi / f(180.0)
                src_lon = lon * pi / f(180.0)
                scxf[istep - 1, knum - 1] = src_lat
                sczf[istep - 1, knum - 1] = src_lon
                periods[istep - 1, knum - 1] = period
                wavetype[istep - 1, knum - 1] = wavetp
                igrt[istep - 1, knum - 1] = veltp
                nsrc1[knum - 1] = istep
                knumo = knum
            else:
                t = line.split()
                lat, lon = f(t[0]), f(t[1])
                istep1 += 1
                rlat = (f(90.0) - lat) * pi / f(180.0)
                rlon = lon * pi / f(180.0)
                rcxf[istep1 - 1, istep - 1, knum - 1] = rlat
                rczf[istep1 - 1, istep - 1, knum - 1] = rlon
                nrc1[istep - 1, knum - 1] = istep1
                vel_obs.append(float(t[2]))
                dist.append(delsph(src_lat, src_lon, rlat, rlon))
    with open(os.path.join(directory, model)) as fh:
        tok = fh.read().split()
    if model == "MOD":
        depz = np.array(tok[:nz], f)
        tok = tok[nz:]
    else:
        depz = load(directory, "MOD")["depz"]
    vels = np.asfortranarray(np.array(tok[:nx * ny * nz], f).reshape(nz, ny, nx).transpose(2, 1, 0))   # vsf(i, j, k)
    return dict(nx=nx, ny=ny, nz=nz, nparpi=(nx - 2) * (ny - 2) * (nz - 1), vels=vels, goxd=goxd, gozd=gozd, dvxd=dvxd, dvzd=dvzd,
                kRc=kRc, kRg=kRg, kLc=kLc, kLg=kLg, tRc=per[0], tRg=per[1], tLc=per[2], tLg=per[3], wavetype=wavetype, igrt=igrt,
                periods=periods, depz=depz, minthk=minthk, scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf, nrc1=nrc1, nsrcsurf1=nsrc1,
                kmax=kmax, nsrcsurf=nsrc, nrcf=nrc, ndata=int(nrc1.sum()), spfra=spfra, ifsyn=ifsyn, noiselevel=noiselevel,
                weight0=weight0, damp=damp, minvel=minvel, maxvel=maxvel, maxiter=maxiter, threshold0=threshold0,
                vel_obs=np.array(vel_obs, f), dist=np.array(dist, f), obst=(np.array(dist, f) / np.array(vel_obs, f)).astype(f))


# ---------------------------------------------------------------------------------------------
# ctypes bindings of the drop-in entries (include/dsurftomo_amd.h), every argument by reference

def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _args(c):
    i32 = lambda v: C.byref(C.c_int(int(v)))
    f32 = lambda v: C.byref(C.c_float(float(v)))
    head = [i32(c["nx"]), i32(c["ny"]), i32(c["nz"]), i32(c["nparpi"]), _ptr(c["vels"])]
    tail = [f32(c["goxd"]), f32(c["gozd"]), f32(c["dvxd"]), f32(c["dvzd"]), i32(c["kRc"]), i32(c["kRg"]), i32(c["kLc"]), i32(c["kLg"]),
            _ptr(c["tRc"]), _ptr(c["tRg"]), _ptr(c["tLc"]), _ptr(c["tLg"]), _ptr(c["wavetype"]), _ptr(c["igrt"]), _ptr(c["periods"]),
            _ptr(c["depz"]), f32(c["minthk"]), _ptr(c["scxf"]), _ptr(c["sczf"]), _ptr(c["rcxf"]), _ptr(c["rczf"]), _ptr(c["nrc1"]),
            _ptr(c["nsrcsurf1"]), i32(c["kmax"]), i32(c["nsrcsurf"]), i32(c["nrcf"])]
    return head, tail


def call_calsurfg(c, capacity=None):
    """dsa_calsurfg on a loaded case -> (dsurf, rw, row, col): COO with 1-based rows (data) and columns"""
    from .engine import load_library
    lib = load_library()
    nd = c["ndata"]
    cap = int(capacity if capacity is not None else c.get("spfra", 1.0) * nd * c["nx"] * c["ny"] * c["nz"])
    iw = np.zeros(cap + 1, np.int32)
    rw = np.zeros(cap, np.float32)
    col = np.zeros(cap, np.int32)
    dsurf = np.zeros(nd, np.float32)
    nar = C.c_int(0)
    head, tail = _args(c)
    lib.dsa_dropin_set_capacity(cap)          # the arrays above: the library refuses to write past them (DSA_ERR_CAPACITY)
    rc = lib.dsa_calsurfg(*head, _ptr(iw), _ptr(rw), _ptr(col), _ptr(dsurf), *tail, C.byref(nar))
    if rc != 0:
        raise RuntimeError("dsa_calsurfg: %s" % lib.dsa_dropin_error().decode())
    n = nar.value
    return dsurf, rw[:n].copy(), iw[1:n + 1].copy(), col[:n].copy()


def call_synthetic(c, noiselevel=0.0):
    from .engine import load_library
    lib = load_library()
    obst = np.zeros(c["ndata"], np.float32)
    head, tail = _args(c)
    rc = lib.dsa_synthetic(*head, _ptr(obst), *tail, C.byref(C.c_float(noiselevel)))
    if rc != 0:
        raise RuntimeError("dsa_synthetic: %s" % lib.dsa_dropin_error().decode())
    return obst


def write_raypaths(path, paths):
    """raypath.out as the reference's (disabled) dump writes it and its scripts/plotpath.py reads it
    (CalSurfG.f90:2276-2283): '# nrp', then nrp lines 'latitude longitude' in degrees.  paths: Engine.ray_paths()."""
    with open(path, "w") as fh:
        for _, pts in paths:
            fh.write(" # %11d\n" % len(pts))
            for lat, lon in pts:
                fh.write("  %14.7f  %14.7f\n" % (lat, lon))
