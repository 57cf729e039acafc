import sys, numpy as np, torch
sys.path.insert(0, '.')
import hmmsort_amd as H
K,N,T=60,4,10_000_000
amps=[(3.0,0.8,0.2),(4.0,0.3,0.2),(2.5,0.6,0.25),(3.5,0.5,0.15)]
temps=np.asfortranarray(np.stack([H.create_spike_template(K,*a) for a in amps],1))
for name,pp in [("bench",[0.003,0.001,0.002,0.0015]),("dense",[0.006,0.004,0.005,0.004])]:
  for seed in (1234, 77):
    y=H.create_signal(T,0.3,pp,temps,seed=seed)
    sm=H.StateMatrix.create(N,K,np.log(pp),False)
    dy=torch.from_numpy(y).cuda(); dx=torch.zeros(T,dtype=torch.int16,device='cuda'); dll=torch.zeros(1,dtype=torch.float64,device='cuda')
    ref=None
    for Hh,B in [(512,512),(256,256),(192,256),(128,256)]:
        H.set_option("halo",Hh); H.set_option("block",B)
        plan=H.Plan(T,sm,temps,0.3); st=torch.cuda.current_stream().cuda_stream
        plan.viterbi(dy,dx,dll,st); d=plan.diagnostics(st)
        x=dx.cpu().numpy()
        if ref is None: ref=x.copy()
        stats=torch.zeros(plan.stats_len(),dtype=torch.float64,device='cuda')
        plan.estep(dy,stats,st); de=plan.diagnostics(st)
        print(name,seed,"H",Hh,"B",B,"vit flags",d[0],"spread %.3g"%d[2],"path diffs vs H512",int((x!=ref).sum()),"| fb flags",de[3],de[5],"err %.2g %.2g"%(de[4],de[6]), flush=True)
        plan.close()
