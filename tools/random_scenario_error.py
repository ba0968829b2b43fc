#!/usr/bin/env python3
"""CPU study (oracle only): integrator error on RandomScenario days against the SciPy-faithful DOPRI5 path and a tight solve."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import t1d_oracle as O
names,tab=O.patient_table()
rs=np.random.RandomState(5)
n=300; K=1440
pid=np.arange(n)%30
cho=np.zeros((K,n))
for j in range(n):
    t,a=O.random_scenario_draw(rs)
    for tt,aa in zip(t,a):
        if tt<K: cho[int(tt),j]=aa
basal0=tab[pid,O.IDX["u2ss"]]*tab[pid,O.IDX["BW"]]/6000.0
pool=[basal0*2*rs.rand(n) for _ in range(8)]
z=np.zeros((120,n))
def run(integ,ns):
    e=O.OracleEnv(pid,sensor="Navigator",normals=z,integrator=integ,n_sub=ns); e.reset()
    out=np.empty((K,n))
    for k in range(K):
        r=e.step(pool[k%8],None,cho[k:k+1]); out[k]=r["bg"]
    return out
ref=run("dopri",4)
tight=run("rk4",64)
print("dopri vs tight: max %.2e"%np.abs(ref-tight).max())
for integ,ns in (("rk4",4),("split",4),("split",6),("split",8),("rk4",8)):
    o=run(integ,ns)
    d=np.abs(o-ref); dt=np.abs(o-tight)
    worst=d.max(0)
    print(integ,ns,"vs dopri max %.2e (p95 over envs %.2e, #envs>1e-3: %d)  vs tight max %.2e"%(d.max(),np.percentile(worst,95),(worst>1e-3).sum(),dt.max()), "worst env",worst.argmax(),names[pid[worst.argmax()]],"meal max",cho[:,worst.argmax()].max())
print("meal sizes: max %.0f, mean of nonzero %.1f"%(cho.max(), cho[cho>0].mean()))
print("---- distribution over 300 envs of max_t |BG - reference|")
for name,o,rf in (("dopri(default) vs tight",ref,tight),("split4 vs dopri",run("split",4),ref),("split4 vs tight",run("split",4),tight),("split8 vs tight",run("split",8),tight)):
    w=np.abs(o-rf).max(0)
    print("%-26s median %.1e p90 %.1e p95 %.1e p99 %.1e max %.1e  frac<=1e-3 %.3f"%(name,np.median(w),np.percentile(w,90),np.percentile(w,95),np.percentile(w,99),w.max(),(w<=1e-3).mean()))
