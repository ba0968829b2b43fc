#!/usr/bin/env python3
"""CPU study (oracle only): integrator error on RandomScenario days with a random basal rate every minute, against the
SciPy-faithful DOPRI5 path and a tight solve.  ~2 minutes for 2 400 env-days."""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from oracle import t1d_oracle as O
names,tab=O.patient_table()
rs=np.random.RandomState(2024)
n=2400; K=1440
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
    t0=time.time()
    e=O.OracleEnv(pid,sensor="Navigator",normals=z,integrator=integ,n_sub=ns); e.reset()
    out=np.empty((K,n))
    for k in range(K):
        out[k]=e.step(pool[k%8],None,cho[k:k+1])["bg"]
    print(integ,ns,"%.0f s"%(time.time()-t0),flush=True)
    return out
ref=run("dopri",4); tight=run("rk4",48)
for integ in ("split","split_adaptive"):
    o=run(integ,4)
    for name,rf in (("vs dopri",ref),("vs tight",tight)):
        w=np.abs(o-rf).max(0)
        print("%-15s %-9s median %.1e p95 %.1e p99 %.1e p99.9 %.1e max %.1e frac<=1e-3 %.4f worst %s"%(integ,name,np.median(w),np.percentile(w,95),np.percentile(w,99),np.percentile(w,99.9),w.max(),(w<=1e-3).mean(),names[pid[w.argmax()]]),flush=True)
w=np.abs(ref-tight).max(0); print("dopri vs tight: median %.1e p99 %.1e max %.1e frac<=1e-3 %.4f"%(np.median(w),np.percentile(w,99),w.max(),(w<=1e-3).mean()))
print("largest meal %.0f g"%cho.max())
