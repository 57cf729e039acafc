import sys, time; sys.path.insert(0, ".")
import numpy as np, hmmsort_amd as H
K,N,T=60,4,10_000_000
temps=np.asfortranarray(np.stack([H.create_spike_template(K,*a) for a in [(3.0,0.8,0.2),(4.0,0.3,0.2),(2.5,0.6,0.25),(3.5,0.5,0.15)]],1)); pp=[0.003,0.001,0.002,0.0015]
y=H.create_signal(T,0.3,pp,temps,seed=1)
sm=H.StateMatrix.create(N,K,np.log(pp),False)
mu=np.asfortranarray(temps*0.9); mu[0,:]=0
H.train_model(y[:200000], sm, mu, 0.35, 2)
t=time.time(); smn,mun,sg=H.train_model(y, sm, mu, 0.35, 8); dt=time.time()-t
print("train_model 8+4 EM steps on %d samples: %.3f s = %.1f ms/step; sigma=%.5f"%(T,dt,dt/12*1e3,sg))
t=time.time(); a=H.train_step(y, sm, mu.copy(order="F"), 0.35); print("one host-buffer train_step: %.1f ms"%((time.time()-t)*1e3))
