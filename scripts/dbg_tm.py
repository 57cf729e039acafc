import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hmmsort_amd as H
from conftest import to_oracle_sm
from oracle import oracle as O; O.build()
def two_templates(H, K):
    return np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2), H.create_spike_template(K, 4.0, 0.3, 0.2)], 1))
temps = two_templates(H, 20)
pp = [0.01, 0.006]
y = H.create_signal(3000, 0.3, pp, temps, seed=5)
sm = H.StateMatrix.create(2, 20, np.log(pp), False)
mu0 = np.asfortranarray(temps * 0.8); mu0[0, :] = 0
osm, omu, osig = to_oracle_sm(O, sm), mu0.copy(order="F"), 0.5
for eng in (H.ENGINE_WAVE, H.ENGINE_RING, H.ENGINE_STRICT):
    H.set_option("engine", eng)
    s, m, sg = sm, mu0.copy(order="F"), 0.5
    o = (to_oracle_sm(O, sm), mu0.copy(order="F"), 0.5)
    for it in range(3):
        s, m, sg = H.train_step(y, s, m, sg)
        o = O.train_step(y, o[0], o[1], o[2])[:3]
        print(eng, it, np.abs(m - o[1]).max(), abs(sg - o[2]), H.get_option("last_escalations"))
H.set_option("engine", 0)
for nst in (2,):
    smn, mu, sig = H.train_model(y, sm, mu0.copy(order="F"), 0.5, nst, None, postprocess=None)
    print("train_model", np.abs(mu - o[1]).max())
p = H.Plan(3000, sm, mu0, 0.5); print(p.info()); 
