"""Deterministic synthetic inputs shared by tests and bench.py.

No RNG library dependence: a 64-bit LCG (seed 20251001) gives the source / receiver positions.
Coordinates follow the reference's conventions: colatitude / longitude in radians
(main.f90:258-274), phase-velocity maps pv[(jj-1)*nx + ii - 1] with the latitude index fastest.
"""
import numpy as np

SEED = 20251001
GOXD, GOZD, DVD = 30.0, 100.0, 0.01
PI32 = np.float32(3.1415926535898)


class LCG:
    def __init__(self, seed=SEED):
        self.s = np.uint64(seed)

    def uniform(self, n):
        out = np.empty(n, np.float64)
        s = int(self.s)
        for i in range(n):
            s = (6364136223846793005 * s + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            out[i] = (s >> 11) / float(1 << 53)
        self.s = np.uint64(s)
        return out


def nprop(nx, gd=8):
    return (nx - 3) * gd + 1


def grid_origin(nx, gd=8):
    """(gox, goz, dnx, dnz) as fp32, the way the engine / reference derive them"""
    f = np.float32
    dvx = f(DVD) * PI32 / f(180.0)
    gox = (f(90.0) - f(GOXD)) * PI32 / f(180.0)
    goz = f(GOZD) * PI32 / f(180.0)
    dn = dvx / f(gd)
    return gox, goz, dn, dn


def medium(nx, kind, period=0):
    """one phase-velocity map (nx*nx,) float64"""
    ny = nx
    i = np.arange(nx, dtype=np.float64)[None, :]
    j = np.arange(ny, dtype=np.float64)[:, None]
    if kind == "homog":
        v = np.full((ny, nx), 3.0)
    elif kind == "smooth":       # config 3/4: (2.8 + 0.05 p) (1 + 0.10 sin(4 pi i/nx) cos(4 pi j/ny))
        v = (2.8 + 0.05 * period) * (1.0 + 0.10 * np.sin(4 * np.pi * i / nx) * np.cos(4 * np.pi * j / ny))
    elif kind == "checker":      # config 5: +-8 %, 16-vertex squares
        v = (2.8 + 0.05 * period) * (1.0 + 0.08 * np.where(((i // 16) + (j // 16)) % 2 == 0, 1.0, -1.0))
    elif kind == "checker4":     # small-grid variant used by parity tests: +-13 %, 4-vertex squares
        v = 3.0 * (1.0 + 0.13 * np.where(((i // 4) + (j // 4)) % 2 == 0, 1.0, -1.0))
    elif kind == "rough":
        r = LCG(SEED + 7 + period).uniform(nx * ny).reshape(ny, nx)
        v = 3.0 * (1.0 + 0.10 * (2 * r - 1))
    elif kind == "wild":         # +-45 % random vertices: many colliding fronts (stress for the exception table of the compact field)
        r = LCG(SEED + 13 + period).uniform(nx * ny).reshape(ny, nx)
        v = 3.0 * (1.0 + 0.45 * (2 * r - 1))
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(v.reshape(-1), np.float64)


def sources(nx, nsrc, gd=8, inner=0.90, seed=SEED):
    """nsrc source positions uniform in the inner fraction of the grid -> (scx, scz) fp32 radians"""
    N = nprop(nx, gd)
    gox, goz, dnx, dnz = grid_origin(nx, gd)
    u = LCG(seed).uniform(2 * nsrc)
    lo = 0.5 * (1.0 - inner) * (N - 1)
    fx = (lo + u[0::2] * inner * (N - 1)).astype(np.float32)
    fz = (lo + u[1::2] * inner * (N - 1)).astype(np.float32)
    return (gox + fx * dnx).astype(np.float32), (goz + fz * dnz).astype(np.float32)


def units(nx, nsrc, nper, nrec, gd=8, seed=SEED):
    """(period, source) units in the reference's order (period slot outer, source inner).
    Receivers of source s: the next `nrec` sources, cyclic. Returns dict of arrays."""
    sx, sz = sources(nx, nsrc, gd, seed=seed)
    map_index = np.repeat(np.arange(nper, dtype=np.int32), nsrc)
    scx = np.tile(sx, nper)
    scz = np.tile(sz, nper)
    idx = (np.arange(nsrc)[:, None] + 1 + np.arange(nrec)[None, :]) % nsrc
    rcx = np.tile(sx[idx].reshape(-1), nper)
    rcz = np.tile(sz[idx].reshape(-1), nper)
    nr = np.full(nsrc * nper, nrec, np.int32)
    return dict(map_index=map_index, scx=scx, scz=scz, nrec=nr, rcx=rcx, rcz=rcz)


def boundary_case(nx=12, ny=11, nz=5, kRc=3, kRg=2, kLc=2, kLg=1, nsrc=4, nrcf=5, dvd=0.05, seed=SEED, deep=False,
                  ragged=True, stations=False):
    """Arguments of one CalSurfG / synthetic call (CalSurfG.f90:939-943) as a dict of numpy arrays in
    Fortran memory order. Period slots run Rc, Rg, Lc, Lg like main.f90:215-245.
    stations: source s sits at the same place in every period slot (a station of a real data set: surfdata.dat lists the same
    stations period after period); default: an own random place per (source, slot)."""
    f = np.float32
    r = LCG(seed + 101)
    kmax = kRc + kRg + kLc + kLg
    depz = (np.array([0.0, 3.0, 7.0, 12.0, 18.0, 26.0, 36.0, 48.0, 62.0])[:nz] * (2.0 if deep else 1.0)).astype(f)
    i = np.arange(nx, dtype=np.float64)[:, None, None]
    j = np.arange(ny, dtype=np.float64)[None, :, None]
    k = np.arange(nz, dtype=np.float64)[None, None, :]
    vels = (2.6 + 0.28 * k) * (1.0 + 0.06 * np.sin(2.2 * np.pi * i / nx + 0.3 * k) * np.cos(1.7 * np.pi * j / ny))
    vels = np.asfortranarray(vels.astype(f))                      # vels(nx, ny, nz)
    tRc = np.array([4.0, 6.0, 9.0, 12.0][:kRc], np.float64)
    tRg = np.array([5.0, 8.0, 11.0][:kRg], np.float64)
    tLc = np.array([4.5, 7.0, 10.0][:kLc], np.float64)
    tLg = np.array([6.0, 9.0][:kLg], np.float64)
    wavetype = np.zeros((nsrc, kmax), np.int32, order="F")
    igrt = np.zeros((nsrc, kmax), np.int32, order="F")
    periods = np.zeros((nsrc, kmax), np.int32, order="F")
    nrc1 = np.zeros((nsrc, kmax), np.int32, order="F")
    nsrcsurf1 = np.zeros(kmax, np.int32)
    scxf = np.zeros((nsrc, kmax), f, order="F")
    sczf = np.zeros((nsrc, kmax), f, order="F")
    rcxf = np.zeros((nrcf, nsrc, kmax), f, order="F")
    rczf = np.zeros((nrcf, nsrc, kmax), f, order="F")
    goxd, gozd = f(24.0), f(121.0)
    # stations inside the propagation grid: latitude goxd - [0, nx-3] dvd, longitude gozd + [0, ny-3] dvd
    slot = 0
    for wt, gr, n in ((2, 0, kRc), (2, 1, kRg), (1, 0, kLc), (1, 1, kLg)):
        for p in range(n):
            ns = nsrc - (slot % 2 if ragged else 0)
            nsrcsurf1[slot] = ns
            for s in range(ns):
                u = r.uniform(2 + 2 * nrcf)
                lat = lambda q: float(goxd) - (0.3 + q * (nx - 3.6)) * dvd
                lon = lambda q: float(gozd) + (0.3 + q * (ny - 3.6)) * dvd
                wavetype[s, slot], igrt[s, slot], periods[s, slot] = wt, gr, p + 1
                scxf[s, slot] = f((90.0 - lat(u[0])) * np.pi / 180.0)
                sczf[s, slot] = f(lon(u[1]) * np.pi / 180.0)
                if stations and slot > 0:
                    scxf[s, slot], sczf[s, slot] = scxf[s, 0], sczf[s, 0]
                nr = nrcf - ((s + slot) % 3 if ragged else 0)
                nrc1[s, slot] = nr
                for q in range(nr):
                    rcxf[q, s, slot] = f((90.0 - lat(u[2 + 2 * q])) * np.pi / 180.0)
                    rczf[q, s, slot] = f(lon(u[3 + 2 * q]) * np.pi / 180.0)
            slot += 1
    ndata = int(nrc1.sum())
    return dict(nx=nx, ny=ny, nz=nz, nparpi=(nx - 2) * (ny - 2) * (nz - 1), vels=vels, goxd=goxd, gozd=gozd,
                dvxd=f(dvd), dvzd=f(dvd), kRc=kRc, kRg=kRg, kLc=kLc, kLg=kLg, tRc=tRc, tRg=tRg, tLc=tLc, tLg=tLg,
                wavetype=wavetype, igrt=igrt, periods=periods, depz=depz, minthk=f((depz[1] - depz[0]) / 3.0),
                scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf, nrc1=nrc1, nsrcsurf1=nsrcsurf1, kmax=kmax,
                nsrcsurf=nsrc, nrcf=nrcf, ndata=ndata)
