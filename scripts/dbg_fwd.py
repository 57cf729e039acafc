"""debug aid: forward values of the wave engine vs the oracle's alpha (single chain)"""
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm
import wave_model as WM
rng = np.random.default_rng(1)
N, K, T = 2, 30, 6000
base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2)]
temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in base], 1))
pp = rng.uniform(1e-3, 4e-3, N) * min(1.0, 60.0 / K) * min(1.0, 4.0 / N)
y = H.create_signal(T, 0.3, pp, temps, seed=1)
sm = H.StateMatrix.create(N, K, np.log(pp), False)
mu = np.asfortranarray(temps * rng.uniform(0.7, 1.2, N)[None, :]); mu[0, :] = 0
alpha = O.forward(y, to_oracle_sm(O, sm), mu, 0.4)
m = WM.Ring(sm, mu, 0.4)
H.set_option("engine", H.ENGINE_WAVE); H.set_option("block", 6016)
plan = H.Plan(T, sm, mu, 0.4)
dy = torch.from_numpy(y).cuda()
stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
plan.estep(dy, stats)
fn = H._lib.lib().hmmsort_plan_debug_array
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
fa0 = np.zeros(T); fn(plan._h, 0, fa0.ctypes.data, T)
fref = np.zeros(T); fn(plan._h, 1, fref.ctypes.data, T)
fv = np.zeros((N, T)); fn(plan._h, 2, fv.ctypes.data, N * T)
t = np.arange(T)
want = alpha[0] - m.A * (t + 1)
err = np.abs(fa0 - want)
print("la0 max err", err.max(), "at", err.argmax(), "first >1e-9:", np.argmax(err > 1e-9) if (err > 1e-9).any() else None)
i = int(np.argmax(err > 1e-9)) if (err > 1e-9).any() else 0
print("around", i, fa0[i-2:i+3], want[i-2:i+3], fref[i-2:i+3])
# ring onset masses: lp_a(t) - R_a(t) + q(y_t; mean(a,1)) = alpha of state (a,1)
Rf, V = WM.ring_scores(y, m)
for a in range(N):
    lp = fref + m.sc[a] + np.log(fv[a])
    d = y - m.mean[a, 0]
    got = lp - (d * d) / m.den
    w = alpha[1 + a * m.L] - m.A * (t + 1)
    e2 = np.abs(got - w)[1:]
    print("ring", a, "onset err max", np.nanmax(e2), "at", np.nanargmax(e2) + 1)
