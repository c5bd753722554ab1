"""Where does a unit the tie census left to the fixed point leave the reference's field?  (round 6)

Stage 1: one call of nsrc x nper units on the medium -- the fixed point with its census (exact_ties = 0) and, when the grid is small
enough to march every unit in seconds, the march (exact_ties = 2, the reference's bits); lists the units the census did not flag, worst
receiver error first.  Stage 2: for the worst of them, the periods of that unit's source alone (one bundle), fields kept, both modes; the
nodes at which the two fields differ, earliest first, with the neighbourhood of the first one in both fields -- the tie (or whatever it
is) that set the difference off sits there.  No oracle involved: exact_ties = 2 is pinned to the reference bit for bit by the GPU tests.
   python3 tools/tie_diagnose.py [nx] [sources] [periods] [medium] [seed offset] [units to look at]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import synth
import fuzz_sources; fuzz_sources.install()      # (DSA_FUZZ_INNER, DSA_FUZZ_SNAP: the sources of a tie_fuzz.py call again)
from dsurftomo_amd.engine import Engine

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nper = int(sys.argv[3]) if len(sys.argv) > 3 else 16
kind = sys.argv[4] if len(sys.argv) > 4 else "checker"
seed_off = int(sys.argv[5]) if len(sys.argv) > 5 else 41
nlook = int(sys.argv[6]) if len(sys.argv) > 6 else 3
pick = os.environ.get("DSA_DIAG_PICK", "flag")      # which units stage 2 looks at: "flag" = not flagged by the default rule, worst first; "notie" = the census saw no tie with an influence at all, yet the times differ
solo = os.environ.get("DSA_DIAG_SOLO", "0") == "1"  # stage 2 unit by unit (k_fim_sorted): the unit's own exception table is then there, and the acceptance times (tau) can be shown
nrec = 32
e = Engine(0)
pv = np.stack([synth.medium(nx, kind, p) for p in range(nper)])
u = synth.units(nx, nsrc, nper, nrec, seed=synth.SEED + seed_off)
n = nsrc * nper
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
N = e.nnx
march_all = N <= 1100
e.set_option("exact_ties", 0); e.plan(**u); t0 = e.solve().reshape(n, nrec)
flags, infl = e.unit_ties()
cnt1, sum1, fr1 = e.unit_tie_sums()
flagged = (flags & 1) != 0
if pick == "notie":
    e.set_option("tie_threshold", 1e-12); e.plan(**u); e.solve(); cnt1, sum1, fr1 = e.unit_tie_sums(); e.set_option("tie_threshold", 2e-5)
    flagged = cnt1 > 0
print(f"N={N} {kind} seed+{seed_off}: {n} units; census flags {int(flagged.sum())} ({100.0 * flagged.mean():.2f} %)", flush=True)
if march_all:
    e.set_option("exact_ties", 2); e.plan(**u); tx = e.solve().reshape(n, nrec)
    d = np.abs(t0.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
    rest = np.nonzero(~flagged)[0]
    order = rest[np.argsort(-d[rest])]
    print(f"units left alone: {rest.size}; beyond 1e-4 s: {int((d[rest] > 1e-4).sum())}; beyond 5e-5: {int((d[rest] > 5e-5).sum())}; worst {d[rest].max() if rest.size else 0:.4g}")
    look = [int(k) for k in order[:nlook]]
else:
    look = [int(k) for k in np.nonzero(~flagged)[0][:nlook]]
    print("units left alone:", np.nonzero(~flagged)[0].tolist())


def sub_units(s):
    """the nper units of source s as a plan of their own"""
    idx = np.array([p * nsrc + s for p in range(nper)])
    rr = (idx[:, None] * nrec + np.arange(nrec)[None, :]).reshape(-1)
    return dict(map_index=u["map_index"][idx], scx=u["scx"][idx], scz=u["scz"][idx], nrec=u["nrec"][idx], rcx=u["rcx"][rr], rcz=u["rcz"][rr])


e.set_option("bundle", 0 if solo else 16 if nper >= 16 else 8 if nper >= 8 else 4)
for kv in [o for o in os.environ.get("DSA_DIAG_OPTS", "").split(",") if o]:      # (more engine options for stage 2, e.g. bundle_refined=2: the refined boxes of the lone bundle in a bundle too)
    e.set_option(kv.split("=")[0], float(kv.split("=")[1]))
for k in look:
    s, p = k % nsrc, k // nsrc
    su = sub_units(s)
    e._L.dsa_keep_fields(e._h, 1)
    e.set_option("exact_ties", 0); e.plan(**su); a0 = e.solve().reshape(nper, nrec)
    fl, inf = e.unit_ties()
    F0 = e.field(p).copy()
    K0 = np.abs(e.debug_field(p, 1)) if solo else None          # acceptance times (unit-by-unit solve only)
    st0 = e.stats()
    R0, S0 = e.refined(p)
    c0, s0, f0 = e.unit_tie_sums()
    e.set_option("exact_ties", 2); e.plan(**su); ax = e.solve().reshape(nper, nrec)
    FX = e.field(p).copy()
    RX, SX = e.refined(p)
    # the refined box's snapshot at the hand-off (the reference's ttnr / nstsr, CalSurfG.f90:1287): where do the two modes' differ?
    both = np.isfinite(R0) & np.isfinite(RX)
    dR = np.where(both, np.abs(R0.astype(np.float64) - RX.astype(np.float64)), 0.0)
    nst = int((S0 != SX).sum())
    print(f"\n--- unit {k}: refined snapshot {R0.shape}: statuses differ at {nst} nodes, times at {int((R0 != RX).sum() - (~both & (np.isinf(R0) == np.isinf(RX))).sum())} (largest {dR.max():.3g} s); "
          f"ties with an influence the census counted in this unit (both stages): {int(c0[p])}, their sum {s0[p]:.3g} s")
    if nst:
        w = np.argwhere(S0 != SX)[:6]
        print("    status differences (ix, iz, fixed point -> march, T fixed point, T march): " + "; ".join(f"({a},{b}) {S0[a, b]}->{SX[a, b]} {R0[a, b]:.7f} {RX[a, b]:.7f}" for a, b in w))
    if (R0 != RX).any():
        w = np.argwhere((R0 != RX) & both)
        if len(w):
            o2 = np.argsort(RX[w[:, 0], w[:, 1]])[:6]
            print("    earliest time differences (ix, iz, T march, T fixed point, status march): " + "; ".join(f"({w[q, 0]},{w[q, 1]}) {RX[w[q, 0], w[q, 1]]:.8f} {R0[w[q, 0], w[q, 1]]:.8f} {SX[w[q, 0], w[q, 1]]}" for q in o2))
    dd = np.abs(a0[p].astype(np.float64) - ax[p].astype(np.float64))
    print(f"\n=== unit {k} (source {s}, period {p}): stage-1 receiver error {d[k] if march_all else float('nan'):.4g} s, census influence {infl[k]:.3g} s; alone in a bundle of {int(st0['bundle_size'])}: "
          f"receiver error {dd.max():.4g} s, census flag {int(fl[p] & 1)} influence {inf[p]:.3g}, same times as in the full call: {bool((a0[p] == t0[k]).all())}")
    diff = np.nonzero(F0 != FX)
    print(f"field: {diff[0].size} of {F0.size} nodes differ; largest |dT| {np.abs(F0.astype(np.float64) - FX.astype(np.float64)).max():.4g} s")
    if not diff[0].size:
        continue
    o = np.argsort(FX[diff], kind="stable")
    ix, iz = diff[0][o], diff[1][o]
    print("earliest differing nodes (ix, iz, T march, T fixed point, dT, dT in ulps):")
    for q in range(min(12, ix.size)):
        a, b = FX[ix[q], iz[q]], F0[ix[q], iz[q]]
        print(f"   ({ix[q]:5d},{iz[q]:5d})  {a:.7f}  {b:.7f}  {float(b) - float(a):+.3e}  {int(np.int64(b.view(np.int32)) - np.int64(a.view(np.int32))):+d}")
    x, z = int(ix[0]), int(iz[0])
    for name, F in (("march", FX), ("fixed point", F0)):
        print(f"  neighbourhood of ({x},{z}) in the {name} field (rows ix-2..ix+2, columns iz-2..iz+2), as float bits relative to the node's own:")
        for dx in range(-3, 4):
            row = []
            for dz in range(-3, 4):
                xx, zz = x + dx, z + dz
                if 0 <= xx < F.shape[0] and 0 <= zz < F.shape[1]:
                    mark = ""
                    if K0 is not None and F is F0 and K0[xx, zz] != F0[xx, zz]:
                        if K0[xx, zz] < 1e-20:
                            mark = f"#{int(round(float(K0[xx, zz]) / 1e-30))}"          # pinned by the serial band march: its accept number (0: alive at the hand-off)
                        else:
                            mark = f"*tau{int(np.int64(K0[xx, zz].view(np.int32)) - np.int64(F[x, z].view(np.int32))):+d}"      # an exceptional node: accepted later than its value
                    row.append(f"{F[xx, zz]:.6f}({int(np.int64(F[xx, zz].view(np.int32)) - np.int64(F[x, z].view(np.int32))):+d}){mark}")
                else:
                    row.append("outside")
            print("     " + "  ".join(row))
    # exact ties of the fixed-point field around the first node: pairs of near neighbours with bit-equal values within 3 nodes
    print("  bit-equal neighbour pairs of the fixed-point field within 3 nodes of it:")
    for xx in range(max(x - 3, 0), min(x + 4, F0.shape[0] - 1)):
        for zz in range(max(z - 3, 0), min(z + 4, F0.shape[1] - 1)):
            if F0[xx, zz] == F0[xx + 1, zz]: print(f"     ({xx},{zz}) == ({xx + 1},{zz}) = {F0[xx, zz]:.7f}")
            if F0[xx, zz] == F0[xx, zz + 1]: print(f"     ({xx},{zz}) == ({xx},{zz + 1}) = {F0[xx, zz]:.7f}")
    # how the difference grows downstream: error quantiles by travel time
    dT = np.abs(F0.astype(np.float64) - FX.astype(np.float64))
    tq = np.quantile(FX, [0.25, 0.5, 0.75, 1.0])
    lo = 0.0
    for hi in tq:
        m = (FX > lo) & (FX <= hi)
        print(f"  T in ({lo:.1f}, {hi:.1f}] s: {int((dT[m] > 0).sum())} nodes differ, largest {dT[m].max():.3g} s")
        lo = hi
e.close()
