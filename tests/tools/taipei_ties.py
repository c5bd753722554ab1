import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'tests','tools'))
import taipei_probe as tp
from dsurftomo_amd import engine as E
taipei = tp.taipei
c = taipei.load()
e = E.Engine(0)
vel = np.ascontiguousarray(c["vels"].T)
e.dispersion_begin(vel, c["depz"], float(c["minthk"]), c["kmax"], c["kmax"])
e.dispersion_run(2, 0, c["tRc"], True, 0, 0)
e.maps_from_dispersion(c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], 8)
maps, sx, sz, nrec, rx, rz = [], [], [], [], [], []
for kk in range(c["kmax"]):
    for s in range(c["nsrcsurf1"][kk]):
        maps.append(c["periods"][s, kk] - 1); sx.append(c["scxf"][s, kk]); sz.append(c["sczf"][s, kk])
        nrec.append(c["nrc1"][s, kk]); rx += list(c["rcxf"][:nrec[-1], s, kk]); rz += list(c["rczf"][:nrec[-1], s, kk])
e._L.dsa_keep_fields(e._h, 1)
e.set_option("exact_ties", 0); e.plan(maps, sx, sz, nrec, rx, rz); t0 = e.solve()
fl, mx = e.unit_ties(); cnt, sm, fr = e.unit_tie_sums()
flagged = np.nonzero(fl & 1)[0]
print("flagged units", flagged.tolist())
print("their sources (sx, sz):", sorted(set((float(sx[k]), float(sz[k])) for k in flagged)))
snap0 = {int(k): e.refined(int(k)) for k in flagged[:6]}
e.set_option("exact_ties", 2); e.plan(maps, sx, sz, nrec, rx, rz); tx = e.solve()
print("fixed point vs march, all units: max |dt| %.3g, times not bit-identical %d" % (np.abs(t0 - tx).max(), int((t0.view(np.uint32) != tx.view(np.uint32)).sum())))
off = np.concatenate([[0], np.cumsum(nrec)])
for k in flagged[:6]:
    k = int(k)
    R0, S0 = snap0[k]; RX, SX = e.refined(k)
    d = np.abs(t0[off[k]:off[k+1]] - tx[off[k]:off[k+1]])
    w = np.argwhere(S0 != SX)
    print("unit", k, "period map", maps[k], "box", S0.shape, ": statuses differ at", len(w), "nodes", w[:4].tolist(), "; receiver times differ", int((d > 0).sum()), "max", float(d.max()) if d.size else 0.0)
e.close()
