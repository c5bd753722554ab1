"""Fuzz the dispersion stage against the oracle: random depth grids, sublayering, velocity columns with
low-velocity zones, short and long periods (where the root search may fail and the reference returns 0)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth, _libs as L
from dsurftomo_amd.engine import Engine
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
r = synth.LCG(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
e = Engine(0)
bad = 0
for k in range(ncases):
    u = r.uniform(8)
    nz = 3 + int(u[0] * 10); nx = 3 + int(u[1] * 5); ny = 3 + int(u[2] * 5)
    depz = np.concatenate([[0.0], np.cumsum(0.3 + 6.0 * r.uniform(nz - 1) * (1 + u[3] * 3))]).astype(np.float32)
    minthk = float(np.float32(1 + int(u[4] * 5)))
    base = np.sort(1.2 + 3.4 * r.uniform(nz))
    vel = (base[:, None, None] * (1.0 + 0.08 * (2 * r.uniform(nz * ny * nx).reshape(nz, ny, nx) - 1))).astype(np.float32)
    if u[5] > 0.5:
        i = 1 + int(u[6] * (nz - 2)); vel[i] *= np.float32(0.8)        # a low-velocity zone
    nper = 2 + int(u[7] * 10)
    t = np.sort(np.concatenate([[0.3 + 0.5 * u[0]], 0.5 + 60.0 * r.uniform(nper - 1) ** 2]))
    for iwave, igr in ((2, 0), (2, 1), (1, 0), (1, 1)):
        ref = L.depthkernel("oracle", vel, depz, minthk, iwave, igr, t)
        try:
            e.dispersion_begin(vel, depz, minthk, nper, nper)
            e.dispersion_run(iwave, igr, t, True, 0, 0)
            dev = e.dispersion_fetch(0, nper, True, 0)
        except Exception as ex:
            print(k, "ERROR", ex); bad += 1; continue
        same = [bool(np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))) for a, b in zip(dev, ref)]
        mx = [float(np.nanmax(np.abs(a - b))) for a, b in zip(dev, ref)]
        zeros = int((ref[0] == 0).sum())
        flag = "" if all(same) else "   <<< max |d| pv %.3g sen %.3g %.3g %.3g" % tuple(mx)
        if flag: bad += 1
        print("%2d nz %2d cols %2d nper %2d minthk %.0f iwave %d igr %d: identical %s, failed roots (cg = 0) %d%s" % (k, nz, nx * ny, nper, minthk, iwave, igr, all(same), zeros, flag), flush=True)
print("combinations with any difference:", bad)
