"""First contact of the multi sweep: one small case against the oracle through the plan API (no fallbacks)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hmmsort_amd as H
import torch
from oracle import oracle as O
from conftest import to_oracle_sm
O.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
rng = np.random.default_rng(5)
temps = np.asfortranarray(np.stack([H.create_spike_template(K, 2.0 + i, 0.3 + 0.2 * i, 0.2) for i in range(N)], 1))
pp = np.array([0.01, 0.006, 0.008, 0.005, 0.007][:N])
SEED = int(sys.argv[4]) if len(sys.argv) > 4 else 3
y = H.create_signal(T, 0.3, pp, temps, seed=SEED)
L = K - 1
for q in range(20):
    t0 = 100 + q * 250; a, b = rng.choice(N, 2, replace=False); d = int(rng.integers(0, L))
    y[t0:t0 + L] += temps[1:, a]; y[t0 + d:t0 + d + L] += temps[1:, b]
sm = H.StateMatrix.create(N, K, np.log(pp), True)
print("states", sm.nstates, "transitions", len(sm.transitions))
xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, 0.3)
st = torch.cuda.current_stream().cuda_stream
dy = torch.from_numpy(y).cuda()
for mode in ("multi", "generic"):
    if mode == "generic": os.environ["HMMSORT_PAIR"] = "0"
    else: os.environ.pop("HMMSORT_PAIR", None)
    H.set_option("block", 1024); H.set_option("halo", 256)
    plan = H.Plan(T, sm, temps, 0.3)
    dx = torch.zeros(T, dtype=torch.int16, device="cuda"); dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    plan.viterbi(dy, dx, dll, st); torch.cuda.synchronize()
    x = dx.cpu().numpy()
    bad = np.nonzero(x != xo)[0]
    print(mode, "diag", plan.diagnostics(st), "mismatches", len(bad), "first", bad[:10], "ll", float(dll.cpu()[0]), llo)
    if len(bad):
        b0 = bad[0]
        print(" x ", x[max(0, b0 - 3):b0 + 12]); print(" xo", xo[max(0, b0 - 3):b0 + 12])
    plan.close()
