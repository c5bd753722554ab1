"""Which local rule would predict the reference's pop order at exact ties?  
The oracle's Fast Marching (bit-pinned to the reference, tests/test_oracle_vs_ref.py) counts, with DSO_TIE_STATS=1, every pop
whose key is bit-equal to the key of a neighbour still in the tree, and for each such tie: whether the popped node had been
inserted into the tree EARLIER than the neighbour (the rule "ties ordered by insertion time"), and on which side of the
neighbour it lies.  A rule is usable only if it predicts (nearly) all of them.  CPU only.
    DSO_TIE_STATS=1 python3 tests/tools/tie_rule_stats.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DSO_TIE_STATS"] = "1"
import _libs as L, synth
O = L.oracle()
O.dso_tie_stats.argtypes = [C.c_void_p, C.c_int]
out = np.zeros(8, np.int64)
for nx, kind in ((131, "smooth"), (131, "rough"), (131, "checker"), (131, "homog"), (259, "checker")):
    g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    pv = synth.medium(nx, kind)
    veln = L.o_gridder(g, pv)
    sx, sz = synth.sources(nx, 12, seed=synth.SEED + 5)
    tot = np.zeros(8, np.int64)
    per = []
    for k in range(len(sx) if nx < 200 else 4):
        O.dso_tie_stats(L.ptr(out), 1)
        L.o_solve(g, pv, veln, sx[k], sz[k])
        O.dso_tie_stats(L.ptr(out), 1)
        tot += out; per.append(int(out[0]))
    n = max(int(tot[0]), 1)
    print("N=%d %-8s: %d sources, exact ties between a popped node and a neighbour in the tree: %d (per field %s); popped one inserted earlier: %.1f %%; "
          "popped one lies at x- %.1f %%, x+ %.1f %%, z- %.1f %%, z+ %.1f %% of the other" %
          (g.nnx, kind, len(per), tot[0], per, 100.0 * tot[1] / n, *(100.0 * tot[2 + q] / n for q in range(4))), flush=True)
