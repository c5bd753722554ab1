"""Fuzz the whole drop-in call against the oracle: random model sizes, period sets, station geometries.
python tests/tools/fuzz_boundary.py [ncases] [seed]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd import engine as E
lib = E.load_library()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = synth.LCG(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
big = len(sys.argv) > 3 and sys.argv[3] == 'big'       # 28..40 vertices per side: rays cross the refined box and long coarse paths
bad = 0
for k in range(ncases):
    u = r.uniform(12)
    kw = dict(nx=(28 if big else 6) + int(u[0] * (12 if big else 16)), ny=(28 if big else 6) + int(u[1] * (12 if big else 16)), nz=3 + int(u[2] * 6), kRc=int(u[3] * 4), kRg=int(u[4] * 3), kLc=int(u[5] * 3), kLg=int(u[6] * 2),
              nsrc=2 + int(u[7] * 6), nrcf=2 + int(u[8] * 6), dvd=0.02 + 0.08 * u[9], seed=int(u[10] * 1e6), deep=bool(u[11] > 0.7))
    if kw["kRc"] + kw["kRg"] + kw["kLc"] + kw["kLg"] == 0: kw["kRc"] = 1
    if os.environ.get('DSA_FUZZ_STATIONS'): kw["stations"] = True        # the same sources at every period slot (with DSA_BUNDLE=4|8|16: bundled solves)
    c = synth.boundary_case(**kw)
    o = L.call_boundary(L.oracle().dso_calsurfg, c)
    try:
        d = L.call_boundary(lib.dsa_calsurfg, c)
    except RuntimeError as ex:
        print(k, kw, "ERROR", ex, lib.dsa_dropin_error()); bad += 1; continue
    so = L.call_boundary(L.oracle().dso_synthetic, c, synthetic=True); sd = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
    dt = float(np.abs(o["dsurf"] - d["dsurf"]).max()); ds = float(np.abs(so - sd).max())
    same = o["nar"] == d["nar"] and np.array_equal(o["iw"], d["iw"]) and np.array_equal(o["col"], d["col"]) and np.array_equal(o["rw"].view(np.uint32), d["rw"].view(np.uint32))
    if not same:
        Go = np.zeros((c["ndata"], c["nparpi"]), np.float32); Gd = np.zeros_like(Go)
        Go[o["iw"] - 1, o["col"] - 1] = o["rw"]; Gd[d["iw"] - 1, d["col"] - 1] = d["rw"]
        gdiff = "G differs: %d entries, max %.3g" % (int((Go.view(np.uint32) != Gd.view(np.uint32)).sum()), float(np.abs(Go - Gd).max()))
    else:
        gdiff = "G identical (%d entries)" % o["nar"]
    flag = "" if (same and dt == 0 and ds == 0) else "   <<<"
    if flag: bad += 1
    print("%2d nx %2d ny %2d nz %d periods %d/%d/%d/%d src %d rec %d dvd %.3f: dsurf max %.3g, synthetic max %.3g, %s%s" %
          (k, kw["nx"], kw["ny"], kw["nz"], kw["kRc"], kw["kRg"], kw["kLc"], kw["kLg"], kw["nsrc"], kw["nrcf"], kw["dvd"], dt, ds, gdiff, flag), flush=True)
print("cases with any difference:", bad, "of", ncases)
