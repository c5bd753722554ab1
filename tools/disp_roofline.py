"""The dispersion stage (k_dispersion + k_depth_kernels; reference surfdisp96.f:223-305, 807-843 under depthkernel, CalSurfG.f90:1-169)
at the headline size: nx = ny = 131, nz = 9, 16 Rayleigh phase periods with depth kernels = 17 161 columns x 55 models x 16 roots.
    python3 tools/disp_roofline.py            -> one JSON line (roots, ms)
    bash tools/collect_pmc.sh disp fp64,busy - -- python3 tools/disp_roofline.py     (FP64 instruction counters of the same run)
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine


def run(e, reps=1, nx=131, nz=9, nper=16):
    c = synth.boundary_case(nx=nx, ny=nx, nz=nz, kRc=1, kRg=0, kLc=0, kLg=0, nsrc=1, nrcf=1)
    vel = np.ascontiguousarray(c["vels"].T)
    t = np.linspace(4.0, 34.0, nper)
    best = None
    for _ in range(reps):
        e.dispersion_begin(vel, c["depz"], float(c["minthk"]), nper, nper)
        e.dispersion_run(2, 0, t, True, 0, 0)
        st = e.stats()
        ms = st["ms_dispersion"]
        best = ms if best is None else min(best, ms)
    curves = st["curves"]
    return {"columns": nx * nx, "models_per_column": 1 + 6 * nz, "periods": nper, "curves": int(curves), "roots": int(curves * nper), "ms": round(best, 3),
            "roots_per_s": round(curves * nper / (best / 1e3), 1)}


if __name__ == "__main__":
    e = Engine(0)
    print(json.dumps(run(e, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
    e.close()
