import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm
N,K,T,seed=8,128,9000,8
rng=np.random.default_rng(seed)
base=[(3.0,0.8,0.2),(4.0,0.3,0.2),(2.5,0.6,0.25),(3.5,0.5,0.15)]
amps=[(base[i%4][0]*(1+0.13*(i//4)),base[i%4][1]+0.03*(i//4),base[i%4][2]) for i in range(N)]
temps=np.asfortranarray(np.stack([H.create_spike_template(K,*a) for a in amps],1))
pp=rng.uniform(1e-3,4e-3,N)*min(1.0,60.0/K)*min(1.0,4.0/N)
y=H.create_signal(T,0.3,pp,temps,seed=seed)
sm=H.StateMatrix.create(N,K,np.log(pp),False)
mu=np.asfortranarray(temps*rng.uniform(0.7,1.2,N)[None,:]); mu[0,:]=0
H.set_option("engine",H.ENGINE_RING)
smn,mun,sgn=H.train_step(y,sm,mu.copy(order='F'),0.4)
print("escalations",H.get_option("last_escalations"))
osmn,omu,osig,olp,opp=O.train_step(y,to_oracle_sm(O,sm),mu.copy(order='F'),0.4)
d=np.abs(mun-omu); print("mu maxabs",d.max(), "at",np.unravel_index(d.argmax(),d.shape), "rel", (d/np.maximum(np.abs(omu),1e-300)).max())
print("sigma",sgn,osig, abs(sgn-osig)/osig)
print("lp",smn.transitions["lp"][:9], olp)
for a in range(N): print(a, "col maxabs", d[:,a].max(), "G-ish count", np.exp(olp[a])*T)

print("---- step 2")
smn2,mun2,sgn2=H.train_step(y,smn,mun.copy(order='F'),sgn)
osmn2,omu2,osig2,olp2,opp2=O.train_step(y,to_oracle_sm(O,smn),mun.copy(order='F'),sgn)
d=np.abs(mun2-omu2); print("mu maxabs",d.max(), "at",np.unravel_index(d.argmax(),d.shape))
bad=~np.isclose(mun2,omu2,rtol=1e-8,atol=1e-11); print("bad entries",bad.sum(), np.argwhere(bad)[:10])
for a in range(N): print(a, "col maxabs", d[:,a].max(), "max|mu|", np.abs(omu2[:,a]).max(), "count", np.exp(olp2[a])*T, "nan", np.isnan(mun2[:,a]).sum(), np.isnan(omu2[:,a]).sum())
print("sigma",sgn2,osig2)
