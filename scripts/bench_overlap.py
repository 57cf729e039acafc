import sys, time; sys.path.insert(0,'.')
import numpy as np, hmmsort_amd as H
for (N,K,T) in [(2,60,100000),(3,60,100000),(4,40,100000),(4,60,20000)]:
    pp=[0.004]*N
    temps=np.asfortranarray(np.stack([H.create_spike_template(K,3.0+0.3*i,0.3+0.1*i,0.2) for i in range(N)],1))
    sm=H.StateMatrix.create(N,K,np.log(pp),True)
    y=H.create_signal(T,0.3,pp,temps,seed=8)
    x,ll=H.viterbi(y,sm,temps,0.3)
    t=time.time(); x,ll=H.viterbi(y,sm,temps,0.3); dt=time.time()-t
    print(N,K,sm.nstates,T,"viterbi %.3fs %.0f samples/s"%(dt,T/dt), flush=True)
