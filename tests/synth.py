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
