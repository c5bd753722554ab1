import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import synth
_src = synth.sources
synth.sources = lambda nx, nsrc, gd=8, inner=0.90, seed=synth.SEED: _src(nx, nsrc, gd, 1.0, seed)
from dsurftomo_amd.engine import Engine
nx,nsrc,nper,nrec=35,1000,16,32
seed=1585
pv=np.stack([synth.medium(nx,"checker4",p) for p in range(nper)])
u=synth.units(nx,nsrc,nper,nrec,seed=synth.SEED+seed)
n=nsrc*nper
import traceback
SRC=int(os.environ.get("DSA_EDGE_SRC","622"))
e=Engine(0)
e.set_maps(nx,nx,synth.GOXD,synth.GOZD,synth.DVD,synth.DVD,pv)
idx=np.array([p*nsrc+SRC for p in range(nper)])
rr=(idx[:,None]*nrec+np.arange(nrec)[None,:]).reshape(-1)
su=dict(map_index=u["map_index"][idx],scx=u["scx"][idx],scz=u["scz"][idx],nrec=u["nrec"][idx],rcx=u["rcx"][rr],rcz=u["rcz"][rr])
gox,goz,dnx,dnz=synth.grid_origin(nx)
print("source",SRC,"fx %.4f fz %.4f"%((u["scx"][SRC]-gox)/dnx,(u["scz"][SRC]-goz)/dnz))
e.set_option("exact_ties",2); e.plan(**su); tx=e.solve().reshape(nper,nrec); print("march times[0,:4]",tx[0,:4])
for b, br in ((16, 1), (0, 1), (16, 2)):
    e.set_option("bundle",b); e.set_option("exact_ties",0); e.set_option("bundle_refined", br)
    e._L.dsa_keep_fields(e._h,1)
    try:
        e.plan(**su); t=e.solve().reshape(nper,nrec); st=e.stats()
        R,S=e.refined(0)
        print("bundle",b,"bundle_refined",br,"times[0,:4]",t[0,:4],"max |dt| vs march",np.abs(t-tx).max(),"rounds_max",st["rounds_max"],"refined box",R.shape,"alive",int((S==0).sum()),"close",int((S>0).sum()),"far",int((S<0).sum()))
    except Exception as ex:
        print("bundle",b,"bundle_refined",br,"ERROR",str(ex)[:200])
        R,S=e.refined(0); print("   refined box",R.shape,"alive",int((S==0).sum()),"close",int((S>0).sum()),"far",int((S<0).sum()))
e.close()
