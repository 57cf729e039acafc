"""debug aid: E-step certificates of the wave engine on chosen shapes (prints geometry + diagnostics)"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch
import hmmsort_amd as H


def case(N, K, T, seed, block=0, halo=0, pp_scale=1.0, pp=None):
    rng = np.random.default_rng(seed)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    if pp is None:
        pp = rng.uniform(2e-4, 3e-4, N) * pp_scale
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * rng.uniform(0.8, 1.1, N)[None, :])
    mu[0, :] = 0
    H.set_option("engine", H.ENGINE_WAVE)
    H.set_option("block", block)
    H.set_option("halo", halo)
    plan = H.Plan(T, sm, mu, 0.4)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    d = plan.diagnostics(st)
    print("N=%d K=%d T=%d block=%d halo=%d -> info %s diag %s" % (N, K, T, block, halo, plan.info(), d), flush=True)
    if d[3] or d[5]:
        import ctypes as C
        rec = (C.c_double * 64)()
        H._lib.lib().hmmsort_plan_debug_record.argtypes = [C.c_void_p, C.c_void_p]
        H._lib.lib().hmmsort_plan_debug_record(plan._h, rec)
        print("   dbg: cg %g dir %g D %r w0 %r d0 %r err %r ringmass %r wb %r bi %g tc %g ne %g" % tuple(rec[:11]))
    plan.close()


if __name__ == "__main__":
    case(16, 200, 14_000, 10)
    case(16, 200, 40_000, 10)
    case(4, 60, 10_000_000, 1234, pp=[0.003, 0.001, 0.002, 0.0015])
