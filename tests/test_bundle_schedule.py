"""The shared round schedule of the bundles (csrc/bundle_kernel.hip) replayed on the CPU (tests/tools/bundle_lab.cpp, with the product's own
solve_node): G coarse problems of ONE source -- its periods -- under one active set routed by member 0.  Every member must come out as its
own solo schedule leaves it, node for node and bit for bit (the fixed point does not depend on the schedule), and equal to the oracle's
field wherever the solo schedule is.  No GPU involved; the GPU side of the same claim is tests/test_gpu_bundles.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _libs as L
import synth
from test_hostcheck import H      # noqa: F401  (fixture: the host build of eikonal_core.h / source_stage.h)

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tools", "bundle_lab.cpp")
SO = os.path.join(HERE, "tools", "libbundle_lab.so")


@pytest.fixture(scope="module")
def lab():
    if L._stale(SO, [SRC, os.path.join(L.ROOT, "dsurftomo_amd", "csrc", "eikonal_core.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-msse2", "-mfpmath=sse", "-o", SO, SRC])
    lib = C.CDLL(SO)
    lib.lab_bundle.argtypes = [L.i32, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.vp, L.i32, L.i32, L.i32, L.vp]
    lib.lab_bundle.restype = C.c_long
    return lib


def maps(nx, kind, G):
    if kind != "mixed":
        return [synth.medium(nx, kind, p) for p in range(G)]
    i = np.arange(nx, dtype=np.float64)[None, :]; j = np.arange(nx, dtype=np.float64)[:, None]
    out = []
    for p in range(G):
        w = p / max(G - 1, 1)
        v = (2.8 + 0.05 * p) * (1.0 + 0.10 * (1 - w) * np.sin(4 * np.pi * i / nx) * np.cos(4 * np.pi * j / nx) + 0.08 * w * np.sin(6 * np.pi * i / nx + 1.0) * np.sin(2 * np.pi * j / nx + 0.5))
        out.append(np.ascontiguousarray(v.reshape(-1), np.float64))
    return out


@pytest.mark.parametrize("nx,kind,G,isrc", [(18, "smooth", 16, 3), (18, "rough", 16, 5), (24, "mixed", 8, 1), (35, "checker4", 4, 2), (24, "rough", 5, 6)])
def test_shared_schedule_reaches_every_members_fixed_point(H, lab, nx, kind, G, isrc):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    N = g.nnx
    sx, sz = synth.sources(nx, 8)
    pvs = maps(nx, kind, G)
    T0 = np.zeros((G, N, N), np.float32); tau0 = np.zeros((G, N, N), np.float32); slow = np.zeros((G, N, N), np.float32)
    ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32); win = np.zeros(G, np.float32)
    for p in range(G):
        assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pvs[p]), sx[isrc], sz[isrc], L.ptr(T0[p]), L.ptr(tau0[p]),
                                   L.ptr(slow[p]), L.ptr(ris), L.ptr(geom)) == 0
        win[p] = np.float32(0.6) * geom[3]
    res = {}
    for rule, pilot in ((0, 0), (2, 0), (2, G - 1), (1, 0)):            # solo; routed by the first / the last member (the product: the first); ANY member
        T = T0.copy(); tau = tau0.copy(); out = np.zeros(16, np.int64)
        assert lab.lab_bundle(G, N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2], L.ptr(win), rule, pilot, 400000, L.ptr(out)) == 0
        res[(rule, pilot)] = (np.abs(T), out.copy())
    solo = res[(0, 0)]
    for key, (T, out) in res.items():
        assert np.array_equal(T.view(np.uint32), solo[0].view(np.uint32)), key
        if key[0]:
            assert out[2] <= 2.2 * solo[1][2]                              # member evaluations: what sharing costs (1.04 .. 1.6 x measured)
    # ... and the solo schedule is the oracle's field where the medium has no exact ties
    if kind in ("smooth", "mixed"):
        veln = L.o_gridder(g, pvs[0])
        o = L.o_solve(g, pvs[0], veln, sx[isrc], sz[isrc])
        assert np.abs(solo[0][0] - o["T"]).max() <= 1e-4
