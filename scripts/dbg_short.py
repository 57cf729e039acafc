"""Forced blocked engine on short signals (shorter than a ring, than a segment, than the warm-up): structured sweeps
against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hmmsort_amd as H
import torch
from oracle import oracle as O
from conftest import to_oracle_sm
O.build()
bad = 0
rng = np.random.default_rng(1)
st = torch.cuda.current_stream().cuda_stream
for N, K in [(2, 40), (2, 7), (3, 12), (4, 10), (3, 30), (5, 6)]:
    shapes = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15), (2.0, 0.4, 0.3)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *shapes[i]) for i in range(N)], 1))
    pp = [0.02, 0.015, 0.01, 0.012, 0.01][:N]
    sm = H.StateMatrix.create(N, K, np.log(pp), True)
    for T in [2, 3, K - 2, K, K + 1, 2 * K, 100, 255, 256, 257, 511, 513, 700, 1500]:
        if T < 2:
            continue
        for sigma in (0.2, 0.45):
            y = H.create_signal(T, sigma, pp, temps, seed=int(rng.integers(1, 1 << 30)))
            xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, sigma)
            H.set_option("engine", H.ENGINE_BLOCKED)
            try:
                plan = H.Plan(T, sm, temps, sigma)
                dy = torch.from_numpy(y).cuda()
                dx = torch.zeros(T, dtype=torch.int16, device="cuda"); dll = torch.zeros(1, dtype=torch.float64, device="cuda")
                plan.viterbi(dy, dx, dll, st); torch.cuda.synchronize()
                d = plan.diagnostics(st); info = plan.info(); plan.close()
                x = dx.cpu().numpy()
                ok = np.array_equal(x, xo) or d[7] > 0 or d[0] > 0
                if not ok:
                    bad += 1
                print("N=%d K=%d T=%d sigma=%.2f: %s diag0 %d ties %d engine %d" % (N, K, T, sigma, "ok" if np.array_equal(x, xo) else ("flagged" if ok else "MISMATCH %d" % int((x != xo).sum())), d[0], d[7], info["engine"]), flush=True)
            except Exception as exc:
                print("N=%d K=%d T=%d: EXC %r" % (N, K, T, exc)); bad += 1
            finally:
                H.set_option("engine", H.ENGINE_AUTO)
print("failures", bad)
sys.exit(1 if bad else 0)
