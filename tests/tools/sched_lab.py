"""Driver of tests/tools/sched_lab.cpp: rounds / evaluations of alternative round schedules on one coarse problem.
   python3 tests/tools/sched_lab.py [nx] [kind] [window list] [mode:param list]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L, synth
H = C.CDLL(os.path.join(ROOT, "tests", "libhostcheck.so"))
H.hc_coarse_problem.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32] + [L.vp] * 5
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
windows = [float(w) for w in (sys.argv[3] if len(sys.argv) > 3 else "1.25").split(",")]
modes = [tuple(int(v) for v in m.split(":")) for m in (sys.argv[4] if len(sys.argv) > 4 else "1:0").split(",")]
lab = C.CDLL(os.path.join(ROOT, "tests", "tools", "libsched_lab.so"))
lab.lab_schedule.argtypes = [L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.f32, L.i32, L.i32, L.i32, L.vp]
lab.lab_schedule.restype = C.c_long
g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
N = g.nnx
pv = synth.medium(nx, kind)
sx, sz = synth.sources(nx, 8)
T0 = np.zeros((N, N), np.float32); tau0 = np.zeros((N, N), np.float32); slow = np.zeros((N, N), np.float32)
ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32)
assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[3], sz[3], L.ptr(T0), L.ptr(tau0), L.ptr(slow), L.ptr(ris), L.ptr(geom)) == 0
ref = None
for w in windows:
    for mode, param in modes:
        T = T0.copy(); tau = tau0.copy(); out = np.zeros(16, np.int64)
        t0 = time.time()
        rc = lab.lab_schedule(N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2], np.float32(w * geom[3]), mode, param, 200000, L.ptr(out))
        same = "first" if ref is None else "identical=%s" % np.array_equal(np.abs(T).view(np.uint32), ref.view(np.uint32))
        if ref is None: ref = np.abs(T)
        print("N=%d %s window %.2f mode %d:%d rc %d: rounds %5d subpasses %6d evals/node %.3f changes/node %.3f ready/round %6.0f listed/round %6.0f max ready %5d freezes %d trips256 %d trips128 %d key-routed %.2f extra-listed/round %.0f sleeping listings %.2f tiles %.2f model_us %.0f | %s (%.1f s)" %
              (N, kind, w, mode, param, rc, out[0], out[4], out[1] / (N * N), out[2] / (N * N), out[5] / max(out[0], 1), out[6] / max(out[0], 1), out[7], out[3], out[8], out[9], out[10] / max(out[6], 1), out[11] / max(out[0], 1), out[12] / max(out[6], 1), out[13] / max(out[14], 1), (out[0] * 19.0 + out[8] * 6.4) / 1000.0, same, time.time() - t0), flush=True)
if os.environ.get("LAB_TAU_STATS"):
    Tf, kf = np.abs(T), np.abs(tau)
    fin = np.isfinite(Tf)
    ne = (Tf != kf) & fin
    pinned = np.signbit(T)
    print("tau != T at %d of %d nodes (%.4f %%), of which pinned %d; tau > T: %d, tau < T: %d; max tau - T %.3g" %
          (ne.sum(), fin.sum(), 100.0 * ne.mean(), (ne & pinned).sum(), ((kf > Tf) & fin).sum(), ((kf < Tf) & fin).sum(), (kf - Tf)[ne & ~pinned].max() if (ne & ~pinned).any() else 0))
