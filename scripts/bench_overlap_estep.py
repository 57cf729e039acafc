"""E-step rate of the blocked engine on an overlap model, device-resident (plan API): python scripts/bench_overlap_estep.py [N K T]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import hmmsort_amd as H
N, K, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 60, 1_000_000)
base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25)]
temps = np.asfortranarray(np.stack([H.create_spike_template(K, *base[i]) for i in range(N)], 1))
pp = [0.012, 0.008, 0.006][:N]
y = H.create_signal(T, 0.3, pp, temps, seed=3)
sm = H.StateMatrix.create(N, K, np.log(pp), True)
H.set_option("engine", H.ENGINE_BLOCKED)
plan = H.Plan(T, sm, temps, 0.3)
dy = torch.from_numpy(y).cuda()
stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
for _ in range(2):
    plan.estep(dy, stats); plan.mstep(stats, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    plan.estep(dy, stats); plan.mstep(stats, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("blocked E-step N=%d K=%d S=%d T=%d: %.2f ms per step = %.1f Msamples/s, diag %s, workspace %.1f GB"
      % (N, K, sm.nstates, T, dt * 1e3, T / dt / 1e6, plan.diagnostics()[3:7], plan.info()["workspace_bytes"] / 1e9))
