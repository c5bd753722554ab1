"""How far does the REFERENCE's own travel-time field depend on which of two exactly tied narrow-band nodes its heap pops first?
The oracle's Fast Marching (bit-pinned to the reference) is run twice on the same input: as the reference does it (the sift-down of
downtree prefers the left child on equal keys, CalSurfG.f90:838-840) and with the right child preferred (DSO_TIE_POLICY=1, a
diagnostic switch of the oracle: an equally valid heap, an equally valid Fast Marching order).  The difference between the two
fields is the part of the reference's answer that is decided by its heap layout; the GPU engine's differences from the reference
at the same sizes (profiles/r02_parity_table.log) are of the same kind and size.  CPU only; run in two processes because the
switch is read once.
    python3 tests/tools/tie_sensitivity.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(131, "smooth", 3), (131, "rough", 0), (131, "checker", 0), (131, "homog", 0), (259, "checker", 1), (515, "checker", 2)]

if len(sys.argv) > 1 and sys.argv[1] == "--solve":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L, synth
    out = {}
    for nx, kind, period in CASES:
        if nx > int(sys.argv[3]):
            continue
        g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
        pv = synth.medium(nx, kind, period)
        veln = L.o_gridder(g, pv)
        N = g.nnx
        sx = np.float32(g.gox + np.float32(0.37 * (N - 1) + 0.3) * g.dnx)
        sz = np.float32(g.goz + np.float32(0.58 * (N - 1) + 0.6) * g.dnz)
        out["%d_%s" % (nx, kind)] = L.o_solve(g, pv, veln, sx, sz)["T"]
    np.savez(sys.argv[2], **out)
    sys.exit(0)

maxnx = sys.argv[1] if len(sys.argv) > 1 else "515"
tmp = "/tmp/tie_sens_%d" % os.getpid()
for pol in ("0", "1"):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--solve", "%s_%s.npz" % (tmp, pol), maxnx], env=dict(os.environ, DSO_TIE_POLICY=pol))
a, b = np.load(tmp + "_0.npz"), np.load(tmp + "_1.npz")
for k in a.files:
    d = np.abs(a[k] - b[k])
    print("N=%4d %-8s: left-child vs right-child tie preference in the reference's heap: field max |dT| %.3g s, nodes beyond 1e-4 s %d (%.4f %%), "
          "nodes not bit-identical %.3f %%, largest time %.1f s" %
          ((int(k.split("_")[0]) - 3) * 8 + 1, k.split("_")[1], d.max(), int((d > 1e-4).sum()), 100.0 * (d > 1e-4).mean(),
           100.0 * (a[k].view(np.uint32) != b[k].view(np.uint32)).mean(), a[k].max()), flush=True)
os.remove(tmp + "_0.npz"); os.remove(tmp + "_1.npz")
