"""Full-size (BASELINE config 2/3: N=4, K=60, 10 M samples) checks through size-independent
properties, plus the strict engine as the bit-exact reference at that size (the CPU oracle would
need ~5 GB and half a minute per decode; it is used at <= 2 M samples elsewhere)."""
import numpy as np
import pytest

from conftest import four_templates

pytestmark = pytest.mark.gpu
T = 10_000_000


@pytest.fixture(scope="module")
def big(H):
    K, N = 60, 4
    temps = four_templates(H, K)
    pp = [0.003, 0.001, 0.002, 0.0015]
    y, onsets = H.create_signal(T, 0.3, pp, temps, seed=1234, return_states=True)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    return dict(K=K, N=N, temps=temps, pp=pp, y=y, onsets=onsets, sm=sm)


@pytest.mark.parametrize("engine", ["wave", "ring"])
def test_ring_decode_equals_strict_decode_at_10M(H, big, engine):
    H.set_option("engine", H.ENGINE_WAVE if engine == "wave" else H.ENGINE_RING)
    x, ll = H.viterbi(big["y"], big["sm"], big["temps"], 0.3)
    assert H.get_option("last_escalations") == 0
    H.set_option("engine", H.ENGINE_STRICT)
    xs, lls = H.viterbi(big["y"], big["sm"], big["temps"], 0.3)
    H.set_option("engine", H.ENGINE_AUTO)
    assert np.array_equal(x, xs)                       # bit-exact path at full size
    assert abs(ll - lls) <= 1e-9 * abs(lls)
    # the decode is a valid path of the model: ring states advance by one, rings start at phase 1
    d = np.diff(x.astype(np.int64))
    ring = x[:-1] > 1
    L = big["K"] - 1
    last = ring & (((x[:-1] - 2) % L) == L - 1)
    assert np.all(d[ring & ~last] == 1)
    nxt = x[1:][last]
    assert np.all((nxt == 1) | (((nxt - 2) % L) == 0))
    # decoded spikes line up with the generated ones (detection is not perfect by design)
    starts = np.nonzero((x[1:] > 1) & (((x[1:] - 2) % L) == 0))[0] + 1
    truth = np.array([t for t, _ in big["onsets"]])
    hit = np.isin(truth, starts) | np.isin(truth + 1, starts) | np.isin(truth - 1, starts)
    assert hit.mean() > 0.9 and len(starts) < 1.1 * len(truth)
    # reconstruct_signal(viterbi) explains the signal up to the noise (reference "Viterbi" test)
    Y = H.reconstruct_signal(x, big["sm"], big["temps"], 0.3)
    q = 1 - np.std(Y - big["y"]) / np.std(big["y"])
    assert abs(np.std(Y - big["y"]) - 0.3) < 0.01 and 0.5 < q < 0.65


def test_estep_properties_at_10M(H, big):
    import torch
    K, N, sm, temps = big["K"], big["N"], big["sm"], big["temps"]
    L, NL = K - 1, N * (K - 1)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(big["y"]).cuda()
    plan = H.Plan(T, sm, temps, 0.3)
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    plan.mstep(stats, out, st)
    diag = plan.diagnostics(st)
    assert diag[3] == 0 and diag[5] == 0 and max(diag[4], diag[6]) < 1e-9   # boundary certificates
    s, o = stats.cpu().numpy(), out.cpu().numpy()
    # every column of gamma is a distribution: total posterior mass = T
    assert abs(s[:NL].sum() + s[3 * NL + N] - T) < 1e-7 * T
    # the true model is (nearly) a fixed point of the EM step
    mu = o[:K * N].reshape((K, N), order="F")
    assert np.abs(mu - temps).max() < 0.02 and abs(o[K * N] - 0.3) < 1e-3
    assert np.allclose(np.exp(o[K * N + 1:K * N + 1 + N]), big["pp"], rtol=0.05)
    # determinism: a second E-step gives bit-identical statistics (no atomics in the reductions)
    stats2 = torch.zeros_like(stats)
    plan.estep(dy, stats2, st)
    torch.cuda.synchronize()
    assert torch.equal(stats, stats2)
    # bind() only shares intermediates: same statistics, same decode
    plan.bind(dy, st)
    stats3 = torch.zeros_like(stats)
    plan.estep(dy, stats3, st)
    plan.unbind()
    torch.cuda.synchronize()
    assert torch.equal(stats, stats3)
    # decode + E-step in one call (three sweeps in one launch) == the two separate calls
    dx1 = torch.zeros(T, dtype=torch.int16, device="cuda")
    dx2 = torch.zeros_like(dx1)
    ll1 = torch.zeros(1, dtype=torch.float64, device="cuda")
    ll2 = torch.zeros_like(ll1)
    stats4 = torch.zeros_like(stats)
    plan.viterbi(dy, dx1, ll1, st)
    plan.decode_estep(dy, dx2, ll2, stats4, st)
    d2 = plan.diagnostics(st)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2) and torch.equal(ll1, ll2) and torch.equal(stats, stats4)
    assert d2[0] == 0 and d2[3] == 0 and d2[5] == 0
    plan.close()


def test_device_resident_em_loop(O, H):
    # EM iterations on a device-resident plan with set_model between steps == host-buffer steps
    import torch
    K, N, Ts = 40, 2, 50_000
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2)], 1))
    pp = [0.004, 0.002]
    y = H.create_signal(Ts, 0.3, pp, temps, seed=3)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * 0.8)
    mu[0, :] = 0
    sig = 0.45
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    plan = H.Plan(Ts, sm, mu, sig)
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
    sm_h, mu_h, sig_h = sm, mu.copy(order="F"), sig
    for _ in range(3):
        plan.estep(dy, stats, st)
        plan.mstep(stats, out, st)
        o = out.cpu().numpy()
        mu_d = np.asfortranarray(o[:K * N].reshape((K, N), order="F"))
        sig_d = float(o[K * N])
        lp_d = o[K * N + 1:K * N + 1 + N]
        sm_d = H.StateMatrix.from_states(sm.states, o[K * N + 1 + N:], K, lp_d, False)
        plan.set_model(sm_d, mu_d, sig_d)
        sm_h, mu_h, sig_h = H.train_step(y, sm_h, mu_h, sig_h)
        assert np.array_equal(mu_d, mu_h) and sig_d == sig_h
        assert np.array_equal(sm_d.transitions["lp"], sm_h.transitions["lp"])
    plan.close()


def test_overlap_decode_blocked_equals_strict_at_2M(H):
    # the reference's Viterbi-test model (test/runtests.jl:17-34: N=2, K=60, overlaps on, 3600
    # states) at 2 M samples: time-parallel blocked engine vs the op-for-op single sweep
    K, To = 60, 2_000_000
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2)], 1))
    pp = [0.003, 0.001]
    sm = H.StateMatrix.create(2, K, np.log(pp), True)
    y = H.create_signal(To, 0.3, pp, temps, seed=77)
    # make sure overlapping spikes occur: add a second template on top of some generated spikes
    rng = np.random.default_rng(5)
    for t0 in rng.integers(1000, To - 1000, 200):
        y[t0:t0 + K] += temps[:, 0]
        y[t0 + 17:t0 + 17 + K] += temps[:, 1]
    H.set_option("engine", H.ENGINE_AUTO)
    x, ll = H.viterbi(y, sm, temps, 0.3)               # AUTO -> blocked for an overlap model
    assert H.get_option("last_escalations") == 0
    H.set_option("engine", H.ENGINE_STRICT)
    xs, lls = H.viterbi(y, sm, temps, 0.3)
    H.set_option("engine", H.ENGINE_AUTO)
    assert np.array_equal(x, xs)
    assert abs(ll - lls) <= 1e-9 * abs(lls)
    assert (x > 1 + 2 * (K - 1)).sum() > 200 * 20      # pair states were decoded
    Y = H.reconstruct_signal(x, sm, temps, 0.3)
    assert abs(np.std(Y - y) - 0.3) < 0.01
