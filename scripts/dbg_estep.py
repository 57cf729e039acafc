"""debug aid: one wave-engine EM step vs the oracle on a small case, several geometries"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm

rng = np.random.default_rng(1)
N, K, T = 2, 30, 6000
base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2)]
temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in base], 1))
pp = rng.uniform(1e-3, 4e-3, N) * min(1.0, 60.0 / K) * min(1.0, 4.0 / N)
y = H.create_signal(T, 0.3, pp, temps, seed=1)
sm = H.StateMatrix.create(N, K, np.log(pp), False)
mu = np.asfortranarray(temps * rng.uniform(0.7, 1.2, N)[None, :]); mu[0, :] = 0
_, omu, osig, olp, opp = O.train_step(y, to_oracle_sm(O, sm), mu.copy(order="F"), 0.4)
H.set_option("engine", H.ENGINE_WAVE)
for blk in (0, 6016, 1024):
    H.set_option("block", blk)
    sm_n, mu_n, sig_n = H.train_step(y, sm, mu.copy(order="F"), 0.4)
    lp = sm_n.transitions["lp"][1:1 + N]
    print("block", blk, "esc", H.get_option("last_escalations"), "dmu", np.abs(mu_n - omu).max(), "dsig", abs(sig_n - osig),
          "dpp", np.nanmax(np.abs(sm_n.pi - opp)), flush=True)
