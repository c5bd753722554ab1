"""Device matrix-vector products against the reference's aprod (aprod.f90:7-60): bit-exact, because every
output element adds its entries in storage order in fp32."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import synth
from dsurftomo_amd import io as taipei
from dsurftomo_amd.engine import Engine, load_library

pytestmark = pytest.mark.gpu


def aprod(fn, mode, m, n, x, y, rw, row, col):
    """the reference's calling convention: iw = [nar, rows, cols]"""
    nar = rw.size
    iw = np.concatenate([[nar], row, col]).astype(np.int32)
    x = np.array(x, np.float32, copy=True)
    y = np.array(y, np.float32, copy=True)
    ib = lambda v: C.byref(C.c_int(int(v)))
    fn(ib(mode), ib(m), ib(n), L.ptr(x), L.ptr(y), ib(iw.size), ib(nar), L.ptr(iw), L.ptr(np.ascontiguousarray(rw, np.float32)))
    return y if mode == 1 else x


def check(e, m, n, rw, row, col, seed):
    r = synth.LCG(seed)
    x = (2.0 * r.uniform(n) - 1.0).astype(np.float32)
    y = (2.0 * r.uniform(m) - 1.0).astype(np.float32)
    e.spmv_load(m, n, rw, row, col)
    fns = [L.oracle().dso_aprod] + ([L.ref().aprod_] if L.ref() is not None else [])
    for mode in (1, 2):
        got = e.spmv(mode, x, y)
        for fn in fns:
            want = aprod(fn, mode, m, n, x, y, rw, row, col)
            assert (got.view(np.uint32) != want.view(np.uint32)).sum() == 0


def test_taipei_matrix_with_weights_and_appended_rows():
    """the matrix of the Taipei forward call, data weights applied and smoothing-like rows appended the way
    main.f90:361-457 does, then both products"""
    c = taipei.load()
    d = L.call_boundary(load_library().dsa_calsurfg, c)
    m0, n = c["ndata"], c["nparpi"]
    r = synth.LCG(3)
    w = (r.uniform(m0) > 0.1).astype(np.float32)
    rw = d["rw"] * w[d["iw"] - 1]
    extra_rows = np.repeat(np.arange(m0 + 1, m0 + 1 + 400, dtype=np.int32), 7)
    extra_cols = (1 + (r.uniform(extra_rows.size) * n).astype(np.int32)).clip(1, n)
    extra_val = np.tile(np.array([6, -1, -1, -1, -1, -1, -1], np.float32) * np.float32(4.0), 400)
    e = Engine(0)
    try:
        check(e, m0 + 400, n, np.concatenate([rw, extra_val]), np.concatenate([d["iw"], extra_rows]), np.concatenate([d["col"], extra_cols]), 11)
    finally:
        e.close()


def test_unsorted_storage_order_and_empty_rows():
    r = synth.LCG(8)
    m, n, nar = 300, 170, 20000
    row = (1 + (r.uniform(nar) * (m - 40)).astype(np.int32)).astype(np.int32)        # the last 40 rows stay empty
    col = (1 + (r.uniform(nar) * n).astype(np.int32)).clip(1, n).astype(np.int32)
    rw = (r.uniform(nar) - 0.5).astype(np.float32)
    e = Engine(0)
    try:
        check(e, m, n, rw, row, col, 5)
        with pytest.raises(Exception):
            e.spmv_load(m, n, rw, row + m, col)
    finally:
        e.close()


def test_dropin_aprod_caches_the_matrix():
    """dsa_aprod with the reference's argument list: same bits as aprod_, also after the matrix changes in place"""
    lib = load_library()
    r = synth.LCG(21)
    m, n, nar = 120, 90, 5000
    row = (1 + (r.uniform(nar) * m).astype(np.int32)).clip(1, m).astype(np.int32)
    col = (1 + (r.uniform(nar) * n).astype(np.int32)).clip(1, n).astype(np.int32)
    rw = (r.uniform(nar) - 0.5).astype(np.float32)
    iw = np.concatenate([[nar], row, col]).astype(np.int32)
    ib = lambda v: C.byref(C.c_int(int(v)))
    for trial in range(3):
        if trial == 2:
            rw *= np.float32(1.5)                         # modified in place: must be noticed
        for mode in (1, 2, 1):
            out = []
            for fn in (lib.dsa_aprod, L.oracle().dso_aprod):
                x = np.linspace(-1, 1, n).astype(np.float32)
                y = np.linspace(2, -2, m).astype(np.float32)
                rc = fn(ib(mode), ib(m), ib(n), L.ptr(x), L.ptr(y), ib(iw.size), ib(nar), L.ptr(iw), L.ptr(rw))
                assert fn is not lib.dsa_aprod or rc == 0, lib.dsa_dropin_error()
                out.append(np.concatenate([x, y]))
            assert (out[0].view(np.uint32) != out[1].view(np.uint32)).sum() == 0
