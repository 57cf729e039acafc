"""debug: where do the per-source-cx kernels (UC = false) go wrong for N > 8?"""
import ctypes as C
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
H.set_option("engine", H.ENGINE_WAVE)
st = torch.cuda.current_stream().cuda_stream
fn = H._lib.lib().hmmsort_plan_debug_array
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]


def arr(plan, which, n):
    a = np.zeros(n)
    H._lib.check(fn(plan._h, which, a.ctypes.data_as(C.c_void_p), n))
    return a


for N, K, T in [(12, 24, 20_000), (13, 24, 20_000), (12, 30, 20_000), (9, 24, 20_000)]:
    rng = np.random.default_rng(4)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, base[i % 4][0] * (1 + 0.15 * (i // 4)),
                                                                 base[i % 4][1] + 0.04 * (i // 4), base[i % 4][2])
                                        for i in range(N)], 1))
    pp = rng.uniform(0.004, 0.012, N) * min(1.0, 4.0 / N)
    y = H.create_signal(T, 0.3, pp, temps, seed=4)
    L = K - 1
    for i in range(6):
        t0 = 2000 + i * (T - 4000) // 6
        a, b = i % N, (i + 1) % N
        y[t0:t0 + L] += 1.5 * temps[1:, a]
        y[t0 + L:t0 + 2 * L] += 1.5 * temps[1:, b]
    sm = _per_source_list(H, H.StateMatrix.create(N, K, np.log(pp), False), N, K, rng)
    osm = to_oracle_sm(O, sm)
    mu = np.asfortranarray(temps * rng.uniform(0.85, 1.1, N)[None, :])
    mu[0, :] = 0
    plan = H.Plan(T, sm, mu, 0.4)
    dy = torch.from_numpy(y).cuda()
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    d = plan.diagnostics(st)
    info = plan.info()
    B = info["block"]
    FA0 = arr(plan, 0, T)
    rho = arr(plan, 3, N * T).reshape(N, T)
    al = O.forward(y, osm, mu, 0.4)
    be = O.backward(y, osm, mu, 0.4)
    ab = al + be
    g = np.logaddexp.reduce(ab, axis=0)
    gam1 = np.exp(ab[1::L][:N] - g[None, :])          # gamma of (a,1)
    dfa = np.diff(FA0)
    dal = np.diff(al[0])
    inner = np.ones(T - 1, bool)
    inner[np.arange(B, T, B) - 1] = False            # chain frames change at chain boundaries
    e_f = np.abs(dfa - dal)[inner]
    e_r = np.abs(rho - gam1)
    print("N=%d K=%d block=%d diag=%s: max |d la0 - d alpha0| = %.3e at %d; max |rho - gamma(a,1)| = %.3e at ring %d t %d; nan rho %d"
          % (N, K, B, d[3:7], np.nanmax(e_f), int(np.nanargmax(e_f)), np.nanmax(e_r), *np.unravel_index(np.nanargmax(e_r), e_r.shape),
             int(np.isnan(rho).sum())))
    plan.close()
