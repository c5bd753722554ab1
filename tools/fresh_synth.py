import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _libs as L, synth
from dsurftomo_amd import engine as E
lib = E.load_library()
c = synth.boundary_case()
for k in range(3):
    try:
        sd = L.call_boundary(lib.dsa_synthetic, c, synthetic=True)
        print("synthetic call", k, "ok", sd[:3], lib.dsa_dropin_error())
    except Exception as ex:
        print("exception", ex)
    print("  err:", lib.dsa_dropin_error())
