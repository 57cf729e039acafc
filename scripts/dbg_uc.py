"""debug: E-step certificates of the N=12, K=24 per-source-cx case"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from conftest import to_oracle_sm  # noqa: E402
from test_gpu_wave_edges import _per_source_list  # noqa: E402

O.build()
N, K, T, seed = 12, 24, 50_000, 4
for perturb in (False, True):
    rng = np.random.default_rng(seed)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, base[i % 4][0] * (1 + 0.15 * (i // 4)),
                                                                 base[i % 4][1] + 0.04 * (i // 4), base[i % 4][2])
                                        for i in range(N)], 1))
    pp = rng.uniform(0.004, 0.012, N) * min(1.0, 4.0 / N)
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    L = K - 1
    for i in range(12):
        t0 = 2000 + i * (T - 4000) // 12
        a, b = i % N, (i + 1 + i // N) % N
        if a == b:
            b = (b + 1) % N
        y[t0:t0 + L] += 1.5 * temps[1:, a]
        y[t0 + L:t0 + 2 * L] += 1.5 * temps[1:, b]
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    if perturb:
        sm = _per_source_list(H, sm, N, K, rng)
    mu = np.asfortranarray(temps * rng.uniform(0.85, 1.1, N)[None, :])
    mu[0, :] = 0
    H.set_option("engine", H.ENGINE_WAVE)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    for halo in (0, 512, 2048):
        H.set_option("halo", halo)
        plan = H.Plan(T, sm, mu, 0.4)
        stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
        out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
        plan.estep(dy, stats, st)
        plan.mstep(stats, out, st)
        d = plan.diagnostics(st)
        o = out.cpu().numpy()
        rec = np.zeros(64)
        import ctypes as C
        fn = H._lib.lib().hmmsort_plan_debug_record
        fn.argtypes = [C.c_void_p, C.c_void_p]
        fn(plan._h, rec.ctypes.data_as(C.c_void_p))
        print("perturb=%s halo=%d info=%s diag=%s sigma=%.6f" % (perturb, halo, plan.info(), d, o[K * N]))
        if d[3] or d[5]:
            print("   first failing certificate:", rec[:11])
        plan.close()
    H.set_option("halo", 0)
    _, omu, osig, olp, _ = O.train_step(y, to_oracle_sm(O, sm), mu.copy(order="F"), 0.4)
    print("   oracle sigma %.6f" % osig)
