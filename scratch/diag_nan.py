import sys; sys.path.insert(0,'.')
import numpy as np, torch
from aircraftoptimalcontrol_amd import batch, problems
pr=problems.step_maneuver(1.0,2e-3)
B=131072
x0=problems.random_x0(B)
bp=batch.BatchProblem(pr.QQt,pr.RRt,pr.QQT,pr.xx_ref,pr.uu_ref,pr.dt)
prm=batch.make_params(stepsize_0=1.0,armijo_maxiters=10)
s=batch.NewtonBatchSolver(bp,B,prm)
s.set_initial_from_x0(x0)
first_bad=np.full(B,-1)
hist=[]
for k in range(10):
    s.iterate(k)
    sc=s.scalars(); hist.append(sc)
    st=sc["status"]
    print(k,"flags:",{f:int(((st&f)!=0).sum()) for f in (1,2,4,8,16)},"mean ntr",sc["ntrials"].mean(),"max",sc["ntrials"].max(), "cost mean", np.nanmean(sc["cost_new"]), "ntr hist", np.bincount(sc["ntrials"],minlength=11).tolist())
    nb=(st&1)!=0
    first_bad[(first_bad<0)&nb]=k
bad=np.nonzero(first_bad>=0)[0]
print("bad count",len(bad), "first_bad hist", np.bincount(first_bad[bad]))
np.savez("gpurun_out/bad.npz", idx=bad, x0=x0[bad], first_bad=first_bad[bad], cost=np.stack([h["cost"][bad] for h in hist],1), step=np.stack([h["stepsize"][bad] for h in hist],1), descent=np.stack([h["descent"][bad] for h in hist],1))
