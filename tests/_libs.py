"""ctypes access to the two checker libraries used by the tests.

  oracle/libdsurf_oracle.so   C restatement (always buildable: gcc)
  oracle/_ref/libdsurf_ref.so the reference's own Fortran + white-box handles (only where
                              /root/reference exists, or where a prebuilt copy travelled)

Test infrastructure only: nothing under dsurftomo_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libdsurf_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdsurf_ref.so")
REFERENCE_SRC = "/root/reference/src"

f32, f64, i32 = C.c_float, C.c_double, C.c_int
vp = C.c_void_p


class Grid(C.Structure):
    _fields_ = [(n, i32) for n in ("nx", "ny", "nvx", "nvz", "gdx", "gdz", "sgdl", "sgs", "nnx", "nnz")] + \
               [(n, f32) for n in ("goxd", "gozd", "dvxd", "dvzd", "gox", "goz", "dvx", "dvz", "dnx", "dnz", "earth")]


class Box(C.Structure):
    _fields_ = [(n, i32) for n in ("vnl", "vnr", "vnt", "vnb", "nnx", "nnz")] + \
               [(n, f32) for n in ("gox", "goz", "dnx", "dnz")]


def ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, n) for n in ("dsurf_oracle.c", "surfdisp_oracle.c", "lsmr_oracle.c", "dsurf_oracle.h")]
    if _stale(ORACLE_SO, srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref():
    """Build oracle/_ref from the reference sources when they are present; else use a prebuilt copy."""
    if os.path.isdir(REFERENCE_SRC):
        if _stale(REF_SO, [os.path.join(ORACLE_DIR, n) for n in ("ref_whitebox.f90", "ref_whitebox_lsmr.f90", "Makefile")]):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(build_oracle())
        L.dso_grid_init.argtypes = [C.POINTER(Grid), i32, i32, f32, f32, f32, f32, i32]
        L.dso_gridder.argtypes = [C.POINTER(Grid), vp, vp]
        L.dso_source_box.argtypes = [C.POINTER(Grid), f32, f32, C.POINTER(Box)]
        L.dso_bsplrefine.argtypes = [C.POINTER(Grid), vp, C.POINTER(Box), vp]
        L.dso_solve_source.argtypes = [C.POINTER(Grid), vp, vp, f32, f32, C.POINTER(Box), vp, vp, vp, vp, vp]
        L.dso_travel_plain.argtypes = [i32, i32, f32, f32, f32, f32, f32, vp, f32, f32, vp]
        L.dso_fouds2_masked.argtypes = [i32, i32, i32, f32, f32, f32, f32, vp, vp, vp, i32, i32]
        L.dso_fouds2_masked.restype = f32
        L.dso_srtimes.argtypes = [C.POINTER(Grid), vp, vp, f32, f32, f32, f32, C.POINTER(f32)]
        L.dso_rpaths.argtypes = [C.POINTER(Grid), C.POINTER(Box), vp, vp, vp, vp, f32, f32, f32, f32, vp,
                                 C.POINTER(i32), C.POINTER(i32)]
        _oracle = L
    return _oracle


def ref():
    """The reference library, or None when it is neither buildable nor prebuilt here."""
    global _ref
    if _ref is None:
        so = build_ref()
        if so is None:
            return None
        L = C.CDLL(so)
        L.ref_wb_init.argtypes = [i32, i32, f32, f32, f32, f32, i32]
        L.ref_wb_solve.argtypes = [f32, f32]
        L.ref_wb_srtimes.argtypes = [f32] * 4
        L.ref_wb_srtimes.restype = f32
        L.ref_wb_rpaths.argtypes = [f32, f32, f32, f32, vp]
        L.ref_wb_rbint.restype = i32
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------
# convenience wrappers (numpy in / numpy out). 2-D fields come back as arrays indexed [ix, iz].

def grid(nx, ny, goxd, gozd, dvxd, dvzd, gd=8):
    g = Grid()
    oracle().dso_grid_init(C.byref(g), nx, ny, goxd, gozd, dvxd, dvzd, gd)
    return g


def o_gridder(g, pv):
    pv = np.ascontiguousarray(pv, np.float64)
    veln = np.zeros((g.nnx, g.nnz), np.float32)
    oracle().dso_gridder(C.byref(g), ptr(pv), ptr(veln))
    return veln


def o_solve(g, pv, veln, x, z):
    pv = np.ascontiguousarray(pv, np.float64)
    b = Box()
    T = np.zeros((g.nnx, g.nnz), np.float32)
    Tr = np.zeros(129 * 129, np.float32)
    Sr = np.zeros(129 * 129, np.int32)
    it = np.zeros((g.nnx, g.nnz), np.float32)
    is_ = np.zeros((g.nnx, g.nnz), np.int32)
    rc = oracle().dso_solve_source(C.byref(g), ptr(pv), ptr(veln), x, z, C.byref(b), ptr(T), ptr(Tr), ptr(Sr),
                                   ptr(it), ptr(is_))
    if rc != 0:
        raise ValueError("source outside grid")
    n = b.nnx * b.nnz
    return dict(box=b, T=T, Tr=Tr[:n].reshape(b.nnx, b.nnz).copy(), Sr=Sr[:n].reshape(b.nnx, b.nnz).copy(),
                inj_t=it, inj_s=is_)


def o_srtimes(g, veln, T, sx, sz, rx, rz):
    t = f32()
    rc = oracle().dso_srtimes(C.byref(g), ptr(veln), ptr(T), sx, sz, rx, rz, C.byref(t))
    if rc != 0:
        raise ValueError("receiver outside grid")
    return np.float32(t.value)


def o_rpaths(g, sol, veln, sx, sz, rx, rz):
    fdm = np.zeros((g.nvx + 2, g.nvz + 2), np.float32)
    rb, ns = i32(0), i32(0)
    Tr = np.ascontiguousarray(sol["Tr"])
    Sr = np.ascontiguousarray(sol["Sr"])
    rc = oracle().dso_rpaths(C.byref(g), C.byref(sol["box"]), ptr(veln), ptr(sol["T"]), ptr(Tr), ptr(Sr),
                             sx, sz, rx, rz, ptr(fdm), C.byref(rb), C.byref(ns))
    if rc != 0:
        raise ValueError("receiver outside grid")
    return fdm, rb.value, ns.value


def o_ray_path(g, sol, veln, sx, sz, rx, rz, cap=1 << 16):
    """points of the ray (latitude, longitude in degrees, the reference's conversion at CalSurfG.f90:2279-2280)"""
    O = oracle()
    O.dso_rpaths_path.argtypes = [C.POINTER(Grid), C.POINTER(Box), vp, vp, vp, vp, f32, f32, f32, f32, vp, C.POINTER(i32), C.POINTER(i32), vp, i32, C.POINTER(i32)]
    fdm = np.zeros((g.nvx + 2, g.nvz + 2), np.float32)
    rb, ns, n = i32(0), i32(0), i32(0)
    path = np.zeros((cap, 2), np.float32)
    rc = O.dso_rpaths_path(C.byref(g), C.byref(sol["box"]), ptr(veln), ptr(sol["T"]), ptr(np.ascontiguousarray(sol["Tr"])),
                           ptr(np.ascontiguousarray(sol["Sr"])), sx, sz, rx, rz, ptr(fdm), C.byref(rb), C.byref(ns), ptr(path), cap, C.byref(n))
    if rc != 0:
        raise ValueError("receiver outside grid")
    assert n.value <= cap
    p = path[:n.value]
    pi = np.float32(3.1415926535898)
    lat = (pi / np.float32(2) - p[:, 0]) * np.float32(180.0) / pi
    lon = p[:, 1] * np.float32(180.0) / pi
    return np.stack([lat, lon], axis=1).astype(np.float32)


class RefWB:
    """White-box session on the reference library (module-global state: one at a time)."""

    def __init__(self, nx, ny, goxd, gozd, dvxd, dvzd, gd=8):
        self.L = ref()
        assert self.L is not None
        self.L.ref_wb_init(nx, ny, goxd, gozd, dvxd, dvzd, gd)
        nnx, nnz = i32(), i32()
        a = [f32() for _ in range(4)]
        self.L.ref_wb_dims(C.byref(nnx), C.byref(nnz), *[C.byref(v) for v in a])
        self.nnx, self.nnz = nnx.value, nnz.value
        self.gox, self.goz, self.dnx, self.dnz = [np.float32(v.value) for v in a]
        self.nvx, self.nvz = nx - 2, ny - 2

    def gridder(self, pv):
        pv = np.ascontiguousarray(pv, np.float64)
        self.L.ref_wb_gridder(ptr(pv))
        out = np.zeros((self.nnx, self.nnz), np.float32)
        self.L.ref_wb_get_veln(ptr(out))
        return out

    def solve(self, x, z):
        self.L.ref_wb_solve(x, z)
        T = np.zeros((self.nnx, self.nnz), np.float32)
        self.L.ref_wb_get_ttn(ptr(T))
        it = np.zeros((self.nnx, self.nnz), np.float32)
        is_ = np.zeros((self.nnx, self.nnz), np.int32)
        self.L.ref_wb_get_injected(ptr(it), ptr(is_))
        rnx, rnz = i32(), i32()
        a = [f32() for _ in range(4)]
        bnd = [i32() for _ in range(4)]
        self.L.ref_wb_refined_dims(C.byref(rnx), C.byref(rnz), *[C.byref(v) for v in a], *[C.byref(v) for v in bnd])
        Tr = np.zeros((rnx.value, rnz.value), np.float32)
        Sr = np.zeros((rnx.value, rnz.value), np.int32)
        self.L.ref_wb_get_refined(ptr(Tr), ptr(Sr))
        return dict(T=T, Tr=Tr, Sr=Sr, inj_t=it, inj_s=is_, rdims=(rnx.value, rnz.value),
                    rgeom=[np.float32(v.value) for v in a], bounds=[v.value for v in bnd])

    def srtimes(self, sx, sz, rx, rz):
        return np.float32(self.L.ref_wb_srtimes(sx, sz, rx, rz))

    def rpaths(self, sx, sz, rx, rz):
        fdm = np.zeros((self.nvx + 2, self.nvz + 2), np.float32)
        self.L.ref_wb_rpaths(sx, sz, rx, rz, ptr(fdm))
        return fdm

    def close(self):
        self.L.ref_wb_release()


def call_boundary(fn, c, synthetic=False, noise=0.0):
    """Call a CalSurfG-shaped entry point (reference `calsurfg_` / oracle `dso_calsurfg` / product
    `dsa_calsurfg`: every argument by reference, CalSurfG.f90:939-943) on synth.boundary_case() data."""
    i32 = lambda v: C.byref(C.c_int(int(v)))
    f32 = lambda v: C.byref(C.c_float(float(v)))
    nd, npar = c["ndata"], c["nparpi"]
    head = [i32(c["nx"]), i32(c["ny"]), i32(c["nz"]), i32(npar), ptr(c["vels"])]
    tail = [f32(c["goxd"]), f32(c["gozd"]), f32(c["dvxd"]), f32(c["dvzd"]), i32(c["kRc"]), i32(c["kRg"]), i32(c["kLc"]), i32(c["kLg"]),
            ptr(c["tRc"]), ptr(c["tRg"]), ptr(c["tLc"]), ptr(c["tLg"]), ptr(c["wavetype"]), ptr(c["igrt"]), ptr(c["periods"]),
            ptr(c["depz"]), f32(c["minthk"]), ptr(c["scxf"]), ptr(c["sczf"]), ptr(c["rcxf"]), ptr(c["rczf"]), ptr(c["nrc1"]),
            ptr(c["nsrcsurf1"]), i32(c["kmax"]), i32(c["nsrcsurf"]), i32(c["nrcf"])]
    own = getattr(fn, "__name__", "").startswith(("dsa_", "dso_"))     # the reference's subroutines return nothing

    def check(rc):
        if own and rc != 0:
            raise RuntimeError("%s returned %d" % (fn.__name__, rc))
    if synthetic:
        obst = np.zeros(nd, np.float32)
        check(fn(*head, ptr(obst), *tail, f32(noise)))
        return obst
    cap = nd * npar + 1
    iw = np.zeros(cap + 1, np.int32)
    rw = np.zeros(cap, np.float32)
    col = np.zeros(cap, np.int32)
    dsurf = np.zeros(nd, np.float32)
    nar = C.c_int(0)
    check(fn(*head, ptr(iw), ptr(rw), ptr(col), ptr(dsurf), *tail, C.byref(nar)))
    n = nar.value
    return dict(dsurf=dsurf, nar=n, rw=rw[:n].copy(), iw=iw[1:n + 1].copy(), col=col[:n].copy())


# ---------------------------------------------------------------------------------------------
# dispersion side: the same call on the oracle (by value) and the reference (Fortran, by reference)

def _ib(v):
    return C.byref(C.c_int(int(v)))


def brocher(vs):
    """vp, rho from vs in fp32 the way the reference's column drivers build their layer model
    (CalSurfG.f90:2340-2346); used to make inputs for surfdisp96 tests"""
    f = np.float32
    vs = np.asarray(vs, f)
    v2 = vs * vs; v3 = v2 * vs; v4 = v3 * vs
    p = f(0.9409) + f(2.0947) * vs - f(0.8206) * v2 + f(0.2683) * v3 - f(0.0251) * v4
    p2 = p * p; p3 = p2 * p; p4 = p3 * p; p5 = p4 * p
    rho = f(1.6612) * p - f(0.4721) * p2 + f(0.0671) * p3 - f(0.0043) * p4 + f(0.000106) * p5
    return p.astype(f), rho.astype(f)


def surfdisp96(which, thk, vpv, vs, rho, iflsph, iwave, mode, igr, t):
    """phase / group velocities for one layered model; which = 'oracle' | 'ref'"""
    kmax = len(t)
    tt = np.zeros(max(kmax, 60), np.float64); tt[:kmax] = t
    cg = np.zeros(max(kmax, 60), np.float64)
    a = [np.ascontiguousarray(x, np.float32).copy() for x in (thk, vpv, vs, rho)]
    if which == "oracle":
        O = oracle()
        O.dso_surfdisp96.argtypes = [vp] * 4 + [i32] * 6 + [vp, vp]
        O.dso_surfdisp96(*[ptr(x) for x in a], len(thk), iflsph, iwave, mode, igr, kmax, ptr(tt), ptr(cg))
    else:
        ref().surfdisp96_(*[ptr(x) for x in a], _ib(len(thk)), _ib(iflsph), _ib(iwave), _ib(mode), _ib(igr), _ib(kmax), ptr(tt), ptr(cg))
    return cg[:kmax].copy()


def depthkernel(which, vel, depz, minthk, iwave, igr, t, kernels=True):
    """vel: (nz, ny, nx) fp32 [= Fortran vel(nx,ny,nz)]. Returns pv (kmax, ny*nx) and, with kernels,
    sen_vs/vp/rho (nz, kmax, ny*nx) -- caldespersion when kernels is False"""
    nz, ny, nx = vel.shape
    kmax = len(t)
    t = np.ascontiguousarray(t, np.float64)
    depz = np.ascontiguousarray(depz, np.float32)
    vel = np.ascontiguousarray(vel, np.float32)
    pv = np.zeros((kmax, ny * nx))
    sen = [np.zeros((nz, kmax, ny * nx)) for _ in range(3)] if kernels else []
    if which == "oracle":
        O = oracle()
        O.dso_caldespersion.argtypes = [i32] * 3 + [vp, vp, i32, i32, i32, vp, vp, f32]
        O.dso_depthkernel.argtypes = [i32] * 3 + [vp] * 5 + [i32, i32, i32, vp, vp, f32]
        if kernels:
            O.dso_depthkernel(nx, ny, nz, ptr(vel), ptr(pv), *[ptr(s) for s in sen], iwave, igr, kmax, ptr(t), ptr(depz), minthk)
        else:
            O.dso_caldespersion(nx, ny, nz, ptr(vel), ptr(pv), iwave, igr, kmax, ptr(t), ptr(depz), minthk)
    else:
        R = ref()
        mt = C.byref(C.c_float(minthk))
        if kernels:
            R.depthkernel_(_ib(nx), _ib(ny), _ib(nz), ptr(vel), ptr(pv), *[ptr(s) for s in sen], _ib(iwave), _ib(igr), _ib(kmax), ptr(t), ptr(depz), mt)
        else:
            R.caldespersion_(_ib(nx), _ib(ny), _ib(nz), ptr(vel), ptr(pv), _ib(iwave), _ib(igr), _ib(kmax), ptr(t), ptr(depz), mt)
    return (pv, *sen) if kernels else pv


def layered_models(n, seed=4):
    """n random layered models (thk, vp, vs, rho, periods) for surfdisp96 tests; some with a mild
    low-velocity zone"""
    import synth
    r = synth.LCG(seed)
    out = []
    for trial in range(n):
        nl = 3 + int(r.uniform(1)[0] * 37)
        thk = (0.5 + 5.5 * r.uniform(nl)).astype(np.float32); thk[-1] = 0
        vs = np.sort(1.5 + 3.1 * r.uniform(nl)).astype(np.float32)
        if trial % 5 == 0:
            vs[1:3] = vs[1:3][::-1].copy()
        vpv, rho = brocher(vs)
        kmax = 3 + int(r.uniform(1)[0] * 27)
        t = np.sort(0.5 + 39.5 * r.uniform(kmax))
        out.append((thk, vpv, vs, rho, t))
    return out
