import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmmsort_amd as H
import torch
from oracle import oracle as O
from conftest import to_oracle_sm, two_templates
O.build()
H.set_option("engine", H.ENGINE_BLOCKED)
st = torch.cuda.current_stream().cuda_stream
for K, T, seed in ((60, 20000, 1234), (20, 6000, 2), (5, 3000, 3), (60, 200000, 4), (64, 50000, 5)):
    temps = two_templates(H, K)
    pp = [0.003, 0.001]
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(2, K, np.log(pp), True)
    plan = H.Plan(T, sm, temps, 0.3)
    dy = torch.from_numpy(y).cuda(); dx = torch.zeros(T, dtype=torch.int16, device="cuda"); dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    plan.viterbi(dy, dx, dll, st)
    d = plan.diagnostics(st)
    x = dx.cpu().numpy(); ll = float(dll.cpu()[0])
    xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, 0.3)
    bad = np.nonzero(x != xo)[0]
    print("K=%d T=%d info=%s diag=%s mismatches=%d first=%s ll rel %.2e" % (K, T, plan.info(), d, len(bad), bad[:5], abs(ll - llo) / abs(llo)))
    plan.close()
