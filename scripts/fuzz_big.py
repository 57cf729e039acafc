"""Long signals: the time-parallel engines (AUTO, escalating host entry) against the op-for-op strict
engine on random models and firing regimes: no-overlap models (ring engine), 0.5-6 M samples, or
with the third argument "overlaps" overlap models (blocked engine), 60-500 k samples.
python scripts/fuzz_big.py [n] [seed] [overlaps]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import hmmsort_amd as H

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
OV = len(sys.argv) > 3 and sys.argv[3] == "overlaps"   # overlap models: blocked engine vs strict
bad = 0
for case in range(n_cases):
    N = int(rng.integers(1, 7)); K = int(rng.integers(17, 80)); T = int(rng.integers(500_000, 6_000_000))
    if OV:
        N = int(rng.integers(2, 5)); K = int(rng.integers(6, {2: 61, 3: 61, 4: 50}[N])); T = int(rng.integers(60_000, 300_000))
        if 1 + N * (K - 1) + N * (N - 1) // 2 * (K - 1) ** 2 > 8000:
            T = min(T, 120_000)   # the strict engine decodes ~20-100 k samples/s at these sizes
    sigma = float(rng.uniform(0.15, 0.6))
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, rng.uniform(1.5, 5), rng.uniform(0.2, 1.0),
                                                                rng.uniform(0.1, 0.4)) for _ in range(N)], 1))
    pp = rng.uniform(3e-4, 6e-3, N) * min(1.0, 40.0 / K) * float(rng.choice([0.3, 1.0, 2.5]))
    y = H.create_signal(T, sigma, pp, temps, seed=int(rng.integers(1, 1 << 30)))
    sm = H.StateMatrix.create(N, K, np.log(pp), OV)
    H.set_option("engine", H.ENGINE_AUTO)
    t = time.time(); x, ll = H.viterbi(y, sm, temps, sigma); t_r = time.time() - t
    esc = H.get_option("last_escalations")
    H.set_option("engine", H.ENGINE_STRICT)
    t = time.time(); xs, lls = H.viterbi(y, sm, temps, sigma); t_s = time.time() - t
    H.set_option("engine", H.ENGINE_AUTO)
    ok = np.array_equal(x, xs) and abs(ll - lls) <= 1e-9 * abs(lls)
    bad += not ok
    print("case %d: N=%d K=%d T=%d sigma=%.2f rate x%.1f -> %s (esc %d; ring %.2fs strict %.2fs)%s"
          % (case, N, K, T, sigma, pp.sum() / 0.01, "ok" if ok else "MISMATCH", esc, t_r, t_s,
             "" if ok else " %d samples differ, first at %d" % ((x != xs).sum(), np.nonzero(x != xs)[0][0])), flush=True)
print("%d cases, %d failures" % (n_cases, bad))
sys.exit(1 if bad else 0)
