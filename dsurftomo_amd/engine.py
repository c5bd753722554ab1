"""Python binding of the engine-level C ABI (include/dsurftomo_amd.h) via ctypes.

Plumbing only: arrays in, arrays out.  There is no CPU implementation behind this class; if the
HIP library is missing or no GPU is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdsurftomo_amd.so")

STAT_NAMES = ("ms_total", "ms_fim_coarse", "ms_fim_refined", "ms_stages", "launches_fim_coarse", "units",
              "rounds_max", "evals_total", "chunk", "rescans", "freezes")

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
    L.dsa_get_dims.argtypes = [_vp, C.POINTER(_i32), C.POINTER(_i32)]
    L.dsa_keep_fields.argtypes = [_vp, _i32]
    L.dsa_get_field.argtypes = [_vp, _i32, _vp]
    L.dsa_get_velocity.argtypes = [_vp, _i32, _vp]
    L.dsa_get_refined.argtypes = [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp]
    L.dsa_get_stats.argtypes = [_vp, _vp]
    L.dsa_debug_field.argtypes = [_vp, _i32, _i32, _vp]
    L.dsa_dropin_error.restype = C.c_char_p
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


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

    def plan(self, map_index, scx, scz, nrec, rcx, rcz):
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
        if rcx.size != self._nrays or rcz.size != self._nrays:
            raise ValueError("receiver arrays must hold sum(nrec) entries")
        self._check(self._L.dsa_plan(self._h, n, _p(map_index), _p(scx), _p(scz), _p(nrec), _p(rcx), _p(rcz)))

    def solve(self, want_times=True):
        out = np.zeros(self._nrays, np.float32) if want_times else None
        self._check(self._L.dsa_solve(self._h, _p(out) if want_times else None))
        return out

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

    def stats(self):
        out = np.zeros(32, np.float64)
        self._check(self._L.dsa_get_stats(self._h, _p(out)))
        d = dict(zip(STAT_NAMES, out[:len(STAT_NAMES)].tolist()))
        d["phase_ticks"] = out[len(STAT_NAMES):len(STAT_NAMES) + 7].tolist()
        return d
