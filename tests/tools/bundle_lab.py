"""Driver of tests/tools/bundle_lab.cpp: what one shared round schedule for the G periods of a source costs.
   python3 tests/tools/bundle_lab.py [nx] [kind] [G] [rules, e.g. 0,1,2:0,2:15,3] [source index]
   kinds: synth.medium's (smooth / checker: the periods' maps are multiples of one pattern; rough: unrelated random maps per period), plus
   'mixed' (two patterns whose weights change with the period: what real phase-velocity maps of one model look like) and 'mixedrough'."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _libs as L, synth
H = C.CDLL(os.path.join(ROOT, "tests", "libhostcheck.so"))
H.hc_coarse_problem.argtypes = [L.i32, L.i32, L.f32, L.f32, L.f32, L.f32, L.i32, L.vp, L.f32, L.f32] + [L.vp] * 5
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 65
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
G = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rules = [tuple(int(v) for v in m.split(":")) for m in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2:0,3").split(",")]
isrc = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lab = C.CDLL(os.path.join(ROOT, "tests", "tools", "libbundle_lab.so"))
lab.lab_bundle.argtypes = [L.i32, L.i32, L.i32, L.vp, L.vp, L.vp, L.vp, L.f32, L.f32, L.f32, L.vp, L.i32, L.i32, L.i32, L.vp]
lab.lab_bundle.restype = C.c_long


def medium(p):
    if kind in ("mixed", "mixedrough"):
        i = np.arange(nx, dtype=np.float64)[None, :]; j = np.arange(nx, dtype=np.float64)[:, None]
        w = p / max(G - 1, 1)
        v = (2.8 + 0.05 * p) * (1.0 + 0.10 * (1 - w) * np.sin(4 * np.pi * i / nx) * np.cos(4 * np.pi * j / nx)
                                + 0.08 * w * np.sin(6 * np.pi * i / nx + 1.0) * np.sin(2 * np.pi * j / nx + 0.5))
        if kind == "mixedrough":
            r = synth.LCG(synth.SEED + 7).uniform(nx * nx).reshape(nx, nx)      # one fine-scale pattern seen by all periods, fading with the period
            v = v * (1.0 + 0.05 * (1 - 0.5 * w) * (2 * r - 1))
        return np.ascontiguousarray(v.reshape(-1), np.float64)
    return synth.medium(nx, kind, p)


g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
N = g.nnx
sx, sz = synth.sources(nx, int(os.environ.get("LAB_NSRC", "8")))
T0 = np.zeros((G, N, N), np.float32); tau0 = np.zeros((G, N, N), np.float32); slow = np.zeros((G, N, N), np.float32)
ris = np.zeros(N, np.float32); geom = np.zeros(4, np.float32); win = np.zeros(G, np.float32)
for p in range(G):
    pv = medium(p)
    assert H.hc_coarse_problem(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8, L.ptr(pv), sx[isrc], sz[isrc], L.ptr(T0[p]), L.ptr(tau0[p]), L.ptr(slow[p]), L.ptr(ris), L.ptr(geom)) == 0
    win[p] = np.float32(float(os.environ.get('LAB_WINDOW', '1.25'))) * geom[3]
ref = None; solo = None
for rule in rules:
    T = T0.copy(); tau = tau0.copy(); out = np.zeros(16, np.int64)
    t0 = time.time()
    rc = lab.lab_bundle(G, N, N, L.ptr(T), L.ptr(tau), L.ptr(slow), L.ptr(ris), geom[0], geom[1], geom[2], L.ptr(win), rule[0], rule[1] if len(rule) > 1 else 0, 400000, L.ptr(out))
    same = "first" if ref is None else "identical=%s" % np.array_equal(np.abs(T).view(np.uint32), ref.view(np.uint32))
    if ref is None: ref = np.abs(T)
    if rule[0] == 0: solo = out.copy()
    rel = "" if solo is None else " | vs solo: member-evals x%.3f, node-evals (listing, addresses) x%.3f of solo's per member, rounds x%.2f of a solo run" % (
        out[2] / solo[2], out[1] / (solo[1] / G), out[0] / (solo[0] / G))
    print("N=%d %s G=%d src %d rule %s rc %d: rounds %d node-evals %d member-evals %d (fill %.3f) ready/round %.0f listed/round %.0f max ready %d freezes %d small changes %.4f of changes%s | %s (%.1f s)" %
          (N, kind, G, isrc, ":".join(map(str, rule)), rc, out[0], out[1], out[2], out[2] / max(out[1] * G, 1) if rule[0] else 1.0, out[3] / max(out[0], 1), out[4] / max(out[0], 1), out[6], out[5], out[8] / max(out[7], 1), rel, same, time.time() - t0), flush=True)
    if ref is not None and not np.array_equal(np.abs(T).view(np.uint32), ref.view(np.uint32)):
        d = np.abs(np.abs(T) - ref); bad = np.abs(T).view(np.uint32) != ref.view(np.uint32)
        print("    differing nodes %d of %d, members %s, max |dT| %.3g, first at %s" % (bad.sum(), bad.size, sorted(set(np.nonzero(bad)[0].tolist())), np.nanmax(np.where(bad, d, 0)), [tuple(int(v) for v in x) for x in np.argwhere(bad)[:4]]))
