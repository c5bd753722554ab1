"""Robustness: a NaN / zero / negative velocity in the model must not hang the device (bounded loops)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _libs as L, synth
from dsurftomo_amd import engine as E
lib = E.load_library()
for what, val in (("nan", np.nan), ("zero", 0.0), ("negative", -1.0), ("huge", 1e30)):
    c = synth.boundary_case()
    v = c["vels"].copy(order="F"); v[5, 4, 2] = val; c["vels"] = v
    t0 = time.time()
    try:
        d = L.call_boundary(lib.dsa_calsurfg, c)
        msg = "returned nar %d, finite dsurf %d/%d" % (d["nar"], int(np.isfinite(d["dsurf"]).sum()), d["dsurf"].size)
    except RuntimeError as ex:
        msg = "error: %s | %s" % (ex, lib.dsa_dropin_error().decode()[:120])
    print("%-8s %.2f s: %s" % (what, time.time() - t0, msg), flush=True)
