"""Batched plans (hmmsort_plan_create_batched): C channels of one recording, each with its own model of the
same shape (the reference sorts one channel per call with that channel's templates, hmmsort.jl:36-104), swept
by one set of launches.  Every channel must come out as its own single-channel plan gives it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def channel_models(H, C, N, K, rng):
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    out = []
    for c in range(C):
        temps = np.asfortranarray(np.stack(
            [H.create_spike_template(K, a * rng.uniform(0.8, 1.2), b, w) for a, b, w in [base[i % 4] for i in range(N)]], 1))
        pp = rng.uniform(0.001, 0.004, N) * (60.0 / K if K > 60 else 1.0)
        out.append((H.StateMatrix.create(N, K, np.log(pp), False), temps, float(rng.uniform(0.25, 0.4)), pp))
    return out


@pytest.mark.parametrize("C,N,K,T", [(3, 4, 60, 200_000), (5, 2, 20, 37_001), (2, 8, 128, 150_000)])
def test_batched_plan_equals_one_plan_per_channel(H, C, N, K, T):
    import torch
    rng = np.random.default_rng(C * 100 + N)
    ms = channel_models(H, C, N, K, rng)
    ys = np.stack([H.create_signal(T, sg, pp, temps, seed=50 + c) for c, (sm, temps, sg, pp) in enumerate(ms)])
    st = torch.cuda.current_stream().cuda_stream
    H.set_option("engine", H.ENGINE_WAVE)
    try:
        # the batched plan starts from channel 0's model everywhere, then every channel gets its own
        plan = H.Plan.batched(T, [ms[0][0]] * C, [ms[0][1]] * C, [ms[0][2]] * C)
        assert plan.channels() == C
        for c, (sm, temps, sg, _) in enumerate(ms):
            plan.set_model_channel(c, sm, temps, sg)
        dy = torch.from_numpy(ys).cuda()
        dx = torch.zeros((C, T), dtype=torch.int16, device="cuda")
        dll = torch.zeros(C, dtype=torch.float64, device="cuda")
        L = plan.stats_len()
        dstats = torch.zeros(C * L, dtype=torch.float64, device="cuda")
        dout = torch.zeros(C * plan.mstep_len(), dtype=torch.float64, device="cuda")
        plan.decode_estep(dy, dx, dll, dstats, st)
        plan.mstep(dstats, dout, st)
        dg = plan.diagnostics(st)
        assert dg[0] == 0 and dg[3] == 0 and dg[5] == 0 and dg[7] == 0, dg
        xb, llb = dx.cpu().numpy(), dll.cpu().numpy()
        ob = dout.cpu().numpy().reshape(C, -1)
        plan.close()
        for c, (sm, temps, sg, _) in enumerate(ms):
            p1 = H.Plan(T, sm, temps, sg)
            d1y = torch.from_numpy(ys[c]).cuda()
            d1x = torch.zeros(T, dtype=torch.int16, device="cuda")
            d1ll = torch.zeros(1, dtype=torch.float64, device="cuda")
            s1 = torch.zeros(p1.stats_len(), dtype=torch.float64, device="cuda")
            o1 = torch.zeros(p1.mstep_len(), dtype=torch.float64, device="cuda")
            p1.decode_estep(d1y, d1x, d1ll, s1, st)
            p1.mstep(s1, o1, st)
            g1 = p1.diagnostics(st)
            assert g1[0] == 0 and g1[3] == 0 and g1[5] == 0
            p1.close()
            assert np.array_equal(xb[c], d1x.cpu().numpy()), "channel %d path" % c
            assert llb[c] == float(d1ll.cpu()[0])
            assert np.allclose(ob[c], o1.cpu().numpy(), rtol=1e-9, atol=1e-12), c
    finally:
        H.set_option("engine", H.ENGINE_AUTO)
