"""Python binding of the engine-level C ABI (include/dsurftomo_amd.h) via ctypes.

Plumbing only: arrays in, arrays out.  There is no CPU implementation behind this class; if the
HIP library is missing or no GPU is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSA_LIB_PATH") or os.path.join(_HERE, "libdsurftomo_amd.so")   # (override: A/B builds)

STAT_NAMES = ("ms_total", "ms_fim_coarse", "ms_fim_refined", "ms_stages", "launches_fim_coarse", "units",
              "rounds_max", "evals_total", "chunk", "rescans", "freezes", "rays", "ray_steps", "rays_clamped",
              "ms_rays", "ms_rows", "nar", "ms_dispersion", "curves", "changes_total", "tie_units", "exact_units", "exact_pops", "ms_exact", "field_slots", "footprint_mb", "bundle_size", "bundles", "bundled_units", "bundle_slots", "bundle_threads", "tie_units_left", "tie_influence_max", "exact_pool", "exact_tiles", "tie_units_strict", "tie_prone_maps", "tie_units_tied", "tie_units_by_scale", "handoffs_replayed")

_f32, _i32, _vp = C.c_float, C.c_int, C.c_void_p
_lib = None


class EngineError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("dsurftomo_amd error %d: %s" % (code, text))
        self.code = code


def load_library():
    """Load the in-tree HIP library; raises if it has not been built (python -m dsurftomo_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s is missing: build it with `python -m dsurftomo_amd.build` "
                                "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.dsa_create.argtypes = [C.POINTER(_vp), _i32]
    L.dsa_destroy.argtypes = [_vp]
    L.dsa_destroy.restype = None
    L.dsa_error_string.argtypes = [_vp]
    L.dsa_error_string.restype = C.c_char_p
    L.dsa_set_memory_budget.argtypes = [_vp, C.c_size_t]
    L.dsa_set_option.argtypes = [_vp, C.c_char_p, C.c_double]
    L.dsa_set_maps.argtypes = [_vp, _i32, _i32, _f32, _f32, _f32, _f32, _i32, _i32, _vp]
    L.dsa_plan.argtypes = [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.dsa_solve.argtypes = [_vp, _vp]
    L.dsa_solve_device.argtypes = [_vp, _vp]
    L.dsa_plan_units.argtypes = [_vp, _i32] + [_vp] * 9
    L.dsa_set_depth_kernels.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]
    L.dsa_solve_rows.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.dsa_dispersion_begin.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp, _f32, _i32, _i32]
    L.dsa_dispersion_run.argtypes = [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32]
    L.dsa_dispersion_copy_maps.argtypes = [_vp, _i32, _i32, _i32]
    L.dsa_dispersion_fetch.argtypes = [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]
    L.dsa_maps_from_dispersion.argtypes = [_vp, _f32, _f32, _f32, _f32, _i32]
    L.dsa_kernels_from_dispersion.argtypes = [_vp]
    L.dsa_get_dims.argtypes = [_vp, C.POINTER(_i32), C.POINTER(_i32)]
    L.dsa_keep_fields.argtypes = [_vp, _i32]
    L.dsa_get_field.argtypes = [_vp, _i32, _vp]
    L.dsa_get_velocity.argtypes = [_vp, _i32, _vp]
    L.dsa_get_refined.argtypes = [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp]
    L.dsa_get_stats.argtypes = [_vp, _vp]
    L.dsa_unit_ties.argtypes = [_vp, _i32, _vp, _vp]
    L.dsa_unit_rounds.argtypes = [_vp, _i32, _vp]
    if hasattr(L, "dsa_unit_tie_sums"):          # (absent from libraries of rounds 1-5: same-box A/B runs against an old build, DSA_LIB_PATH)
        L.dsa_unit_tie_sums.argtypes = [_vp, _i32, _vp, _vp, _vp]
    L.dsa_debug_counters.argtypes = [_vp, _vp]
    L.dsa_ray_paths.argtypes = [_vp, _vp, _vp, _vp]
    L.dsa_spmv_load.argtypes = [_vp, _i32, _i32, C.c_longlong, _vp, _vp, _vp]
    L.dsa_spmv.argtypes = [_vp, _i32, _vp, _vp]
    L.dsa_lsmr.argtypes = [_vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _vp] + [_vp] * 7
    L.dsa_debug_field.argtypes = [_vp, _i32, _i32, _vp]
    L.dsa_selfcheck_divisions.argtypes = [C.c_ulonglong, _i32, _vp, _vp]
    L.dsa_dropin_error.restype = C.c_char_p
    L.dsa_dropin_set_capacity.argtypes = [C.c_longlong]
    L.dsa_aprod_invalidate.argtypes = []
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def selfcheck_divisions(seed, millions, exponents8):
    """device self-check of the hand-expanded divisions (include/dsurftomo_amd.h: dsa_selfcheck_divisions): (fp64 pairs, fp64 quotients that
    differ from the compiler's division bitwise, fp32 pairs, fp32 quotients that differ)"""
    L = load_library()
    ex = np.ascontiguousarray(exponents8, np.int32)
    out = np.zeros(4, np.uint64)
    rc = L.dsa_selfcheck_divisions(int(seed), int(millions), _p(ex), _p(out))
    if rc != 0:
        raise EngineError("dsa_selfcheck_divisions failed (%d)" % rc)
    return tuple(int(v) for v in out)


class Engine:
    def __init__(self, device=0):
        self._L = load_library()
        h = _vp()
        rc = self._L.dsa_create(C.byref(h), int(device))
        if rc != 0:
            raise EngineError(rc, self._L.dsa_error_string(None).decode())
        self._h = h
        self.nnx = self.nnz = 0
        self._nrays = 0
        self._ndata = 0
        self._disp = (0, 0, 0)

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsa_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._L.dsa_error_string(self._h).decode())

    def set_memory_budget(self, nbytes):
        self._check(self._L.dsa_set_memory_budget(self._h, int(nbytes)))

    def set_option(self, name, value):
        self._check(self._L.dsa_set_option(self._h, name.encode(), float(value)))

    def set_maps(self, nx, ny, goxd, gozd, dvxd, dvzd, pv, dicing=8):
        """pv: (nmaps, nx*ny) float64, latitude index fastest inside a map."""
        pv = np.ascontiguousarray(pv, np.float64).reshape(-1, nx * ny)
        self._check(self._L.dsa_set_maps(self._h, nx, ny, goxd, gozd, dvxd, dvzd, dicing, pv.shape[0], _p(pv)))
        a, b = _i32(), _i32()
        self._check(self._L.dsa_get_dims(self._h, C.byref(a), C.byref(b)))
        self.nnx, self.nnz = a.value, b.value

    def plan(self, map_index, scx, scz, nrec, rcx, rcz, mode=None, sen_slot=None, data_first=None):
        """mode / sen_slot / data_first: optional per-unit arrays (see dsa_plan_units)"""
        map_index = np.ascontiguousarray(map_index, np.int32)
        scx = np.ascontiguousarray(scx, np.float32)
        scz = np.ascontiguousarray(scz, np.float32)
        nrec = np.ascontiguousarray(nrec, np.int32)
        rcx = np.ascontiguousarray(rcx, np.float32)
        rcz = np.ascontiguousarray(rcz, np.float32)
        n = map_index.size
        if not (scx.size == n and scz.size == n and nrec.size == n):
            raise ValueError("per-unit arrays differ in length")
        self._nrays = int(nrec.sum())
        self._nrec_of_plan = nrec
        if rcx.size != self._nrays or rcz.size != self._nrays:
            raise ValueError("receiver arrays must hold sum(nrec) entries")
        opt = [None if a is None else np.ascontiguousarray(a, np.int32) for a in (mode, sen_slot, data_first)]
        if any(a is not None and a.size != n for a in opt):
            raise ValueError("per-unit arrays differ in length")
        self._ndata = self._nrays if opt[2] is None else (int((opt[2] + nrec).max()) if n else 0)
        self._check(self._L.dsa_plan_units(self._h, n, _p(map_index), _p(scx), _p(scz), _p(nrec), _p(rcx), _p(rcz),
                                           *[None if a is None else _p(a) for a in opt]))

    def solve(self, want_times=True):
        out = np.zeros(self._ndata, np.float32) if want_times else None
        self._check(self._L.dsa_solve(self._h, _p(out) if want_times else None))
        return out

    def solve_device(self, device_ptr):
        """solve; the receiver times (ndata float32) go to `device_ptr`, memory of this engine's GPU (e.g. tensor.data_ptr())"""
        self._check(self._L.dsa_solve_device(self._h, C.c_void_p(int(device_ptr))))

    @property
    def ndata(self):
        return self._ndata

    def set_depth_kernels(self, vels, depz, sen_vs, sen_vp, sen_rho):
        """vels: (nz, ny, nx) fp32 [Fortran vels(nx,ny,nz)]; sen_*: (nz, kmax, ny*nx) fp64 [Fortran (nx*ny, kmax, nz)]"""
        vels = np.ascontiguousarray(vels, np.float32)
        depz = np.ascontiguousarray(depz, np.float32)
        sen = [np.ascontiguousarray(a, np.float64) for a in (sen_vs, sen_vp, sen_rho)]
        nz, kmax = sen[0].shape[0], sen[0].shape[1]
        self._check(self._L.dsa_set_depth_kernels(self._h, nz, kmax, _p(vels), _p(depz), *[_p(a) for a in sen]))

    def solve_rows(self, capacity):
        """receiver times plus Frechet rows as COO (rw, row, col), rows / columns 1-based"""
        out = np.zeros(self._ndata, np.float32)
        rw = np.zeros(capacity, np.float32)
        iw = np.zeros(capacity, np.int32)
        col = np.zeros(capacity, np.int32)
        nar = C.c_longlong(0)
        self._check(self._L.dsa_solve_rows(self._h, _p(out), _p(rw), _p(iw), _p(col), capacity, C.byref(nar)))
        n = nar.value
        return out, rw[:n].copy(), iw[:n].copy(), col[:n].copy()

    def solve_rows_device(self):
        """receiver times; the Frechet rows stay on the device (set_option('rows_on_device', 1) before it): returns (times, number of entries)"""
        out = np.zeros(self._ndata, np.float32)
        nar = C.c_longlong(0)
        self._check(self._L.dsa_solve_rows(self._h, _p(out), None, None, None, C.c_longlong(1 << 62), C.byref(nar)))
        return out, nar.value

    def ray_paths(self, cap):
        """paths of the rays traced by the last solve_rows (set_option('ray_path_cap', cap) before it): list of
        (datum, points) with points an (n, 2) array of (latitude, longitude) in degrees, receiver first, source last --
        what the reference's disabled dump writes to raypath.out (CalSurfG.f90:2276-2283)"""
        nr = int(self.stats()["rays"])
        datum = np.zeros(nr, np.int32); npts = np.zeros(nr, np.int32); pts = np.zeros((nr, cap, 2), np.float32)
        self._check(self._L.dsa_ray_paths(self._h, _p(datum), _p(npts), _p(pts)))
        if (npts > cap).any():
            raise RuntimeError("ray_paths: a ray has %d points, more than ray_path_cap = %d" % (int(npts.max()), cap))
        return [(int(datum[r]), pts[r, :npts[r]].copy()) for r in range(nr)]

    # ---- dispersion stage ------------------------------------------------------------------------
    def dispersion_begin(self, vels, depz, minthk, kmax_total, nmaps_total):
        """vels: (nz, ny, nx) fp32"""
        vels = np.ascontiguousarray(vels, np.float32)
        nz, ny, nx = vels.shape
        self._disp = (nx, ny, nz)
        self._check(self._L.dsa_dispersion_begin(self._h, nx, ny, nz, _p(vels), _p(np.ascontiguousarray(depz, np.float32)),
                                                 float(minthk), int(kmax_total), int(nmaps_total)))

    def dispersion_run(self, iwave, igr, t, kernels, sen_slot=0, map_first=0):
        t = np.ascontiguousarray(t, np.float64)
        self._check(self._L.dsa_dispersion_run(self._h, iwave, igr, t.size, _p(t), int(bool(kernels)), sen_slot, map_first))

    def dispersion_fetch(self, map_first, nper, kernels=False, sen_slot=0):
        nx, ny, nz = self._disp
        pv = np.zeros((nper, nx * ny))
        sen = [np.zeros((nz, nper, nx * ny)) for _ in range(3)] if kernels else [None] * 3
        self._check(self._L.dsa_dispersion_fetch(self._h, map_first, nper, _p(pv), int(bool(kernels)), sen_slot,
                                                 *[None if a is None else _p(a) for a in sen]))
        return (pv, *sen) if kernels else pv

    def maps_from_dispersion(self, goxd, gozd, dvxd, dvzd, dicing=8):
        self._check(self._L.dsa_maps_from_dispersion(self._h, goxd, gozd, dvxd, dvzd, dicing))
        a, b = _i32(), _i32()
        self._check(self._L.dsa_get_dims(self._h, C.byref(a), C.byref(b)))
        self.nnx, self.nnz = a.value, b.value

    def kernels_from_dispersion(self):
        self._check(self._L.dsa_kernels_from_dispersion(self._h))

    # ---- matrix-vector products of the inversion step (reference aprod) ---------------------------
    def spmv_load(self, m, n, rw, row, col):
        """COO matrix with 1-based row / col indices"""
        rw = np.ascontiguousarray(rw, np.float32)
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        self._mn = (int(m), int(n))
        self._check(self._L.dsa_spmv_load(self._h, int(m), int(n), rw.size, _p(rw), _p(row), _p(col)))

    def spmv(self, mode, x, y):
        """mode 1: returns y + A x; mode 2: returns x + A^T y (fp32, the reference's accumulation order)"""
        x = np.array(x, np.float32, copy=True)
        y = np.array(y, np.float32, copy=True)
        assert x.size == self._mn[1] and y.size == self._mn[0]
        self._check(self._L.dsa_spmv(self._h, int(mode), _p(x), _p(y)))
        return y if mode == 1 else x

    def lsmr(self, b, damp, atol=1e-6, btol=1e-6, conlim=100.0, itnlim=400, local_size=10):
        """LSMR (reference lsmrModule.f90:36, arguments of main.f90:470-489) on the matrix of the last spmv_load;
        returns dict(x, istop, itn, normA, condA, normr, normAr, normx)"""
        b = np.ascontiguousarray(b, np.float32)
        assert b.size == self._mn[0]
        x = np.zeros(self._mn[1], np.float32)
        ii = [C.c_int(-1), C.c_int(-1)]
        ff = [C.c_float(0.0) for _ in range(5)]
        self._check(self._L.dsa_lsmr(self._h, _p(b), damp, atol, btol, conlim, int(itnlim), int(local_size), _p(x),
                                     *[C.byref(v) for v in ii], *[C.byref(v) for v in ff]))
        names = ("normA", "condA", "normr", "normAr", "normx")
        return dict(x=x, istop=ii[0].value, itn=ii[1].value, **{k: np.float32(v.value) for k, v in zip(names, ff)})

    def traveltimes(self, map_index, scx, scz, nrec, rcx, rcz):
        self.plan(map_index, scx, scz, nrec, rcx, rcz)
        return self.solve()

    def field(self, unit):
        """coarse travel-time field of a unit of the last chunk, indexed [ix, iz]"""
        out = np.zeros((self.nnx, self.nnz), np.float32)
        self._check(self._L.dsa_get_field(self._h, int(unit), _p(out)))
        return out

    def velocity(self, m):
        out = np.zeros((self.nnx, self.nnz), np.float32)
        self._check(self._L.dsa_get_velocity(self._h, int(m), _p(out)))
        return out

    def refined(self, unit):
        t = np.zeros(129 * 129, np.float32)
        s = np.zeros(129 * 129, np.int8)
        a, b = _i32(), _i32()
        self._check(self._L.dsa_get_refined(self._h, int(unit), C.byref(a), C.byref(b), _p(t), _p(s)))
        n = a.value * b.value
        return t[:n].reshape(a.value, b.value).copy(), s[:n].reshape(a.value, b.value).copy()

    def debug_field(self, unit, which):
        """raw device state of a resident unit (see dsa_debug_field); coarse fields come back [ix, iz]"""
        out = np.zeros((self.nnx, self.nnz) if which < 2 else (129 * 129,), np.float32)
        self._check(self._L.dsa_debug_field(self._h, int(unit), int(which), _p(out)))
        return out

    def debug_counters(self):
        out = np.zeros(24, np.float64)
        self._check(self._L.dsa_debug_counters(self._h, _p(out)))
        return out

    def unit_ties(self):
        """per unit of the last solve: flags (bit 0 met an exact tie, bit 1 solved by the literal march) and the largest tie influence (s)"""
        n = len(self._nrec_of_plan)
        fl = np.zeros(n, np.int32); inf = np.zeros(n, np.float32)
        self._check(self._L.dsa_unit_ties(self._h, n, _p(fl), _p(inf)))
        return fl, inf

    def unit_tie_sums(self):
        """per unit of the last solve: ties with an influence, the sum of their influences (s), cycles frozen by the unit or its bundle"""
        n = len(self._nrec_of_plan)
        cnt = np.zeros(n, np.int32); sm = np.zeros(n, np.float32); fr = np.zeros(n, np.int32)
        self._check(self._L.dsa_unit_tie_sums(self._h, n, _p(cnt), _p(sm), _p(fr)))
        return cnt, sm, fr

    def unit_rounds(self):
        """rounds of each unit's coarse solve in the last solve"""
        n = len(self._nrec_of_plan)
        r = np.zeros(n, np.int32)
        self._check(self._L.dsa_unit_rounds(self._h, n, _p(r)))
        return r

    def stats(self):
        out = np.zeros(64, np.float64)
        self._check(self._L.dsa_get_stats(self._h, _p(out)))
        d = dict(zip(STAT_NAMES, out[:len(STAT_NAMES)].tolist()))
        d["phase_ticks"] = out[len(STAT_NAMES):len(STAT_NAMES) + 8].tolist()
        return d
