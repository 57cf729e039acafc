"""Randomised parity sweep of the EM loop: train_model (device-resident session) against three
oracle EM steps, from perturbed and from random starts.  python scripts/fuzz_em.py [n] [seed]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.integers(1, 7)); K = int(rng.integers(17, 70))
    T = int(rng.integers(2000, 30000))
    while T * (1 + N * (K - 1)) > 5_000_000:
        T //= 2
    sigma = float(rng.uniform(0.2, 0.5))
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, rng.uniform(2, 5), rng.uniform(0.2, 1.0),
                                                                rng.uniform(0.1, 0.4)) for _ in range(N)], 1))
    pp = rng.uniform(1e-3, 6e-3, N) * min(1.0, 30.0 / K)
    y = H.create_signal(T, sigma, pp, temps, seed=int(rng.integers(1, 1 << 30)))
    random_start = rng.random() < 0.4
    if random_start:   # baumwelch.jl:311-322
        s0 = float(np.std(y, ddof=1))
        lp = np.log(np.full(N, 2.0 ** (-3 * K / 2)))
        mu = np.ones((K, N), order="F")
        for i in range(N):
            mu[:, i] = H.create_spike_template(K, 3 * s0 * rng.random(), 0.5 + 0.1 * rng.standard_normal(), 1.5 * rng.random())
    else:
        s0 = sigma * float(rng.uniform(0.9, 1.4))
        lp = np.log(pp * rng.uniform(0.5, 2.0, N))
        mu = np.asfortranarray(temps * rng.uniform(0.7, 1.3, N)[None, :])
    mu[0, :] = 0
    sm = H.StateMatrix.create(N, K, lp, False)
    tag = "case %d: N=%d K=%d T=%d %s" % (case, N, K, T, "random start" if random_start else "perturbed")
    try:
        sm_n, mu_n, sig_n = H.train_model(y, sm, mu.copy(order="F"), s0, 2, postprocess=None)   # the plain loop: 2 + 1 steps
        osm, omu, osig = to_oracle_sm(O, sm), mu.copy(order="F"), s0
        for _ in range(3):
            osm, omu, osig, _, _ = O.train_step(y, osm, omu, osig)
        fin = np.isfinite(omu)
        ok = (np.array_equal(np.isfinite(mu_n), fin) and np.allclose(mu_n[fin], omu[fin], rtol=1e-6, atol=1e-9)
              and (abs(sig_n - osig) <= 1e-6 * osig or not np.isfinite(osig))
              and len(sm_n.transitions) == len(osm.val)
              and np.allclose(sm_n.transitions["lp"], osm.val, rtol=1e-6, atol=1e-9))
        msg = "ok" if ok else "MISMATCH max|dmu| %.2e dsig %.2e" % (np.nanmax(np.abs(mu_n - omu)), abs(sig_n - osig))
    except Exception as exc:  # noqa: BLE001
        ok, msg = False, "EXCEPTION %r" % (exc,)
    bad += not ok
    print(tag, "->", msg, flush=True)
print("%d cases, %d failures, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
