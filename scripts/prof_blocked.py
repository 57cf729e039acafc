"""Device-resident decode of an overlap model through the plan API (blocked engine); prints the rate.
usage: python scripts/prof_blocked.py [N K T reps]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hmmsort_amd as H
from hmmsort_amd import device

N, K, T, reps = (int(a) for a in (sys.argv[1:5] + ["2", "60", "4000000", "3"][len(sys.argv) - 1:]))
pp = [0.004] * N
temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0 + 0.3 * i, 0.3 + 0.1 * i, 0.2)
                                    for i in range(N)], 1))
sm = H.StateMatrix.create(N, K, np.log(pp), True)
y = H.create_signal(T, 0.3, pp, temps, seed=8)
H.set_option("engine", H.ENGINE_BLOCKED)
import os
H.set_option("block", int(os.environ.get("HMMSORT_BLOCK", "0")))
H.set_option("halo", int(os.environ.get("HMMSORT_HALO", "0")))
p = device.Plan(T, sm, temps, 0.3)
print("plan", p.info(), flush=True)
dy = torch.from_numpy(y).cuda()
dx = torch.zeros(T, dtype=torch.int16, device="cuda")
dll = torch.zeros(1, dtype=torch.float64, device="cuda")
p.viterbi(dy, dx, dll); torch.cuda.synchronize()
t = time.time()
for _ in range(reps):
    p.viterbi(dy, dx, dll)
torch.cuda.synchronize()
dt = (time.time() - t) / reps
print(f"N={N} K={K} S={sm.nstates} T={T}: {dt*1e3:.2f} ms/decode, {T/dt/1e6:.1f} Msamples/s, diag={p.diagnostics()[:3]}",
      flush=True)
