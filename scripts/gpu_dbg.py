import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mpc_oracle as O, orc
from model_predictive_control_amd import engine, variants, workload
name='dsr_diff'; B=2048
v=variants.get(name,chain=1); ov=O.variant(name,chain=1)
courses=workload.make_courses(16); inst=workload.make_instances(courses,B,seed=11)
xref=workload.reference_window(courses,inst,5); x0=inst['state'][:,:3].copy()
rng=np.random.default_rng(5)
uref=np.zeros((B,4,5)); uref[:,0]=rng.uniform(-0.4,0.4,(B,5)); uref[:,2]=uref[:,0]; uref[:,1]=rng.uniform(-0.5,0.5,(B,5)); uref[:,3]=uref[:,1]
xbar=orc.rollout_batch(ov,x0,uref); xbar[:,:2]+=rng.normal(0,0.05,(B,2,6))
out=engine.linear_mpc_batch(v,xref,xbar,x0,uref)
st=out['status'].cpu().numpy(); it=out['iters'].cpu().numpy()
bad=np.where(st>1)[0]
print('bad',bad,st[bad],it[bad])
print('iters hist',np.bincount(it))
np.savez('gpurun_out/dbg_diff.npz',bad=bad,xref=xref[bad],xbar=xbar[bad],x0=x0[bad],uref=uref[bad],u=out['u'].cpu().numpy()[bad])
# the instance alone
for b in bad:
    o=engine.linear_mpc_batch(v,xref[b:b+1],xbar[b:b+1],x0[b:b+1],uref[b:b+1])
    print('alone',o['status'].cpu().numpy(),o['iters'].cpu().numpy())
