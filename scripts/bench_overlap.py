"""Decode rate of overlap-resolving models (reference types.jl:78-90) through the host-buffer
entry point, blocked engine vs strict engine.  Run on the GPU box: python scripts/bench_overlap.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import hmmsort_amd as H

cases = [(2, 60, 100000), (2, 60, 2000000), (3, 60, 400000), (4, 40, 400000), (4, 60, 100000)]
for (N, K, T) in cases:
    pp = [0.004] * N
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0 + 0.3 * i, 0.3 + 0.1 * i, 0.2)
                                        for i in range(N)], 1))
    sm = H.StateMatrix.create(N, K, np.log(pp), True)
    y = H.create_signal(T, 0.3, pp, temps, seed=8)
    res = {}
    for name, eng in (("blocked", H.ENGINE_BLOCKED), ("strict", H.ENGINE_STRICT)):
        if eng == H.ENGINE_STRICT and T > 100000:
            continue
        H.set_option("engine", eng)
        x, ll = H.viterbi(y, sm, temps, 0.3)
        t = time.time(); x, ll = H.viterbi(y, sm, temps, 0.3); dt = time.time() - t
        res[name] = (x, ll)
        print(f"N={N} K={K} S={sm.nstates} T={T} {name}: {dt:.3f}s {T/dt/1e6:.3f} Msamples/s "
              f"esc={H.get_option('last_escalations')}", flush=True)
    if len(res) == 2:
        print("   same path:", np.array_equal(res["blocked"][0], res["strict"][0]),
              " ll rel diff: %.2e" % (abs(res["blocked"][1] - res["strict"][1]) / abs(res["strict"][1])), flush=True)
H.set_option("engine", H.ENGINE_AUTO)
