"""The refined box of ONE unit as the hand-off leaves it (snapshot times and statuses: the reference's ttnr / nstsr, CalSurfG.f90:1287), from the fixed point
(unit by unit and in a bundle of the source's periods) and from the march, with the nodes where they differ and the neighbourhood of the earliest one.
   python3 tools/refined_dump.py [nx] [medium] [seed offset] [source] [period]      (DSA_FUZZ_SNAP / DSA_FUZZ_INNER as in tools/tie_fuzz.py)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import synth, fuzz_sources; fuzz_sources.install()
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]); kind = sys.argv[2]; seed = int(sys.argv[3]); s = int(sys.argv[4]); p = int(sys.argv[5])
nsrc, nper, nrec = 1000, 16, 32
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, q) for q in range(nper)])
u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + seed)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
idx = np.array([q * nsrc + s for q in range(nper)])
rr = (idx[:, None] * nrec + np.arange(nrec)[None, :]).reshape(-1)
su = dict(map_index=u["map_index"][idx], scx=u["scx"][idx], scz=u["scz"][idx], nrec=u["nrec"][idx], rcx=u["rcx"][rr], rcz=u["rcz"][rr])
gox, goz, dnx, dnz = synth.grid_origin(nx, 8)
print("source at node coordinates (x, z):", (su["scx"][0] - gox) / dnx, (su["scz"][0] - goz) / dnz, " grid", e.nnx)
e._L.dsa_keep_fields(e._h, 1)
out = {}
for mode, b in ((0, 0), (0, 16), (2, 16)):
    e.set_option("bundle", b); e.set_option("exact_ties", mode); e.plan(**su); e.solve()
    R, S = e.refined(p)
    out[(mode, b)] = (R.copy(), S.copy())
RX, SX = out[(2, 16)]
for key in ((0, 0), (0, 16)):
    R0, S0 = out[key]
    d = np.argwhere((R0 != RX) & np.isfinite(R0) & np.isfinite(RX))
    print("fixed point", "unit by unit" if key[1] == 0 else "in a bundle", ": snapshot", R0.shape, "differs from the march's at", len(d), "nodes; statuses differ at", int((S0 != SX).sum()))
    if not len(d):
        continue
    o = np.argsort(RX[d[:, 0], d[:, 1]])
    for q in o[:8]:
        a, b = d[q]
        print(f"    ({a},{b}) march {RX[a, b]:.9g} fixed point {R0[a, b]:.9g} ({int(R0[a, b].view(np.int32)) - int(RX[a, b].view(np.int32)):+d} ulps) status {SX[a, b]}")
    a, b = d[o[0]]
    for nm, R, S in (("march", RX, SX), ("fixed point", R0, S0)):
        print(f"  {nm} around ({a},{b}): rows ix {a - 4}..{a + 4}, columns iz {b - 4}..{b + 4}; value(status)")
        for x in range(max(a - 4, 0), min(a + 5, R.shape[0])):
            print("     ", x, "  ".join(f"{R[x, z]:.9g}({S[x, z]:+d})" for z in range(max(b - 4, 0), min(b + 5, R.shape[1]))))
e.close()
