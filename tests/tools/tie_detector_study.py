"""Would a scan of the converged field (final-state ties only) flag fewer units than the kernel's inline tie detector (any evaluation)?
   CPU study on the product's own solver (tests/hostcheck.cpp: hc_tie_study).  python3 tests/tools/tie_detector_study.py [nx] [kind] [nsrc]"""
import ctypes as C, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L, synth
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 67
kind = sys.argv[2] if len(sys.argv) > 2 else "rough"
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
so = os.path.join(ROOT, "tests", "libhostcheck.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-msse2", "-mfpmath=sse", "-shared", "-o", so, os.path.join(ROOT, "tests", "hostcheck.cpp"), "-lm"])
H = C.CDLL(so)
H.hc_coarse_problem.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32] + [L.vp] * 5
H.hc_tie_study.argtypes = [L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.f32, L.f32, L.vp]
g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
N = g.nnx
pv = synth.medium(nx, kind)
sx, sz = synth.sources(nx, nsrc, seed=synth.SEED + 3)
for thr in (0.0, 2e-5):
    ni = nf = 0
    for k in range(nsrc):
        T = np.zeros((N, N), np.float32); tau = np.zeros((N, N), np.float32); slow = np.zeros((N, N), np.float32); ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32)
        assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[k], sz[k], L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), L.ptr(geom)) == 0
        out = np.zeros(4)
        assert H.hc_tie_study(N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2], np.float32(1.25 * geom[3]), thr, L.ptr(out)) == 0
        ni += out[0] > thr; nf += out[1] > thr
        if thr > 0: print("  source %2d: inline max %.3g s (%d evaluations)  final-state max %.3g s (%d nodes)" % (k, out[0], out[2], out[1], out[3]))
    print("N=%d %s threshold %g: units flagged by the inline detector %d of %d, by a final-state scan %d" % (N, kind, thr, ni, nsrc, nf), flush=True)
