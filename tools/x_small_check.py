import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import _libs as L, synth
from dsurftomo_amd.engine import Engine
from test_gpu_parity import FRAC, positions
nx, kind, gd = 18, "homog", 8
srcs = positions(nx, gd, FRAC)
g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, gd)
pv = synth.medium(nx, kind); veln = L.o_gridder(g, pv)
e = Engine(0); e.set_option("exact_ties", 2)
e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv, dicing=gd)
n = len(srcs); N = g.nnx
rcx = np.array([[srcs[(i + 3) % n][0]] for i, s in enumerate(srcs)], np.float32); rcz = np.array([[srcs[(i + 3) % n][1]] for i, s in enumerate(srcs)], np.float32)
t = e.traveltimes(np.zeros(n, np.int32), [s[0] for s in srcs], [s[1] for s in srcs], np.full(n, 1, np.int32), rcx.reshape(-1), rcz.reshape(-1))
for u, src in enumerate(srcs[:4]):
    o = L.o_solve(g, pv, veln, src[0], src[1]); T = e.field(u)
    bad = T.view(np.uint32) != o["T"].view(np.uint32)
    Tr, Sr = e.refined(u)
    cls_o = np.sign(o["Sr"]).clip(-1, 1)
    print(os.environ.get("DSA_LIB_PATH", "default"), "unit", u, "coarse nodes differing", int(bad.sum()), "max", float(np.abs(T - o["T"])[np.isfinite(T)].max()), "| refined status differing", int((cls_o != Sr).sum()),
          "refined alive values differing", int((Tr[cls_o == 0].view(np.uint32) != o["Tr"][cls_o == 0].view(np.uint32)).sum()), "first bad", np.argwhere(bad)[:3].tolist())
# earliest differing node of the refined stage for the first bad unit
for u, src in enumerate(srcs):
    o = L.o_solve(g, pv, veln, src[0], src[1])
    Tr, Sr = e.refined(u)
    cls_o = np.sign(o["Sr"]).clip(-1, 1)
    al = (cls_o == 0) & (Sr == 0)
    bad = al & (Tr.view(np.uint32) != o["Tr"].view(np.uint32))
    if not bad.any() and (cls_o == Sr).all(): continue
    print("unit", u, "source", src, "refined shape", Tr.shape, "box", o.get("box"))
    if bad.any():
        idx = np.argwhere(bad); k = np.argmin(o["Tr"][bad]); ix, iz = idx[k]
        print(" earliest differing alive refined node (ix, iz)", int(ix), int(iz), "oracle T %.9g device %.9g" % (o["Tr"][ix, iz], Tr[ix, iz]))
        for dx, dz in ((-2,0),(-1,0),(1,0),(2,0),(0,-2),(0,-1),(0,1),(0,2)):
            x, z = ix + dx, iz + dz
            if 0 <= x < Tr.shape[0] and 0 <= z < Tr.shape[1]:
                print("   nb (%+d,%+d): oracle T %.9g st %d | device T %.9g st %d" % (dx, dz, o["Tr"][x, z], cls_o[x, z], Tr[x, z], Sr[x, z]))
    break
