import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx,nsrc,nper,nrec=131,1000,16,32
kind=sys.argv[1] if len(sys.argv)>1 else "smooth"
e=Engine(0)
pv=np.stack([synth.medium(nx,kind,p) for p in range(nper)])
u=synth.units(nx,nsrc,nper,nrec)
e.set_maps(nx,nx,synth.GOXD,synth.GOZD,synth.DVD,synth.DVD,pv)
e.set_option("exact_ties",0); e.plan(**u); e.solve(); e.solve()
st=e.stats()
os.environ["DSA_DEBUG_CAND"]="1"
cand,_,_=e.unit_tie_sums()
del os.environ["DSA_DEBUG_CAND"]
cnt,sm,fr=e.unit_tie_sums()
print(kind, "coarse ms", st["ms_fim_coarse"], "candidates per bundle: median", np.median(cand), "mean", cand.mean(), "max", cand.max(), "bundles over 135k:", (cand>135000).sum()//16, " ties per unit median", np.median(cnt))
e.close()
