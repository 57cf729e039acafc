"""GPU parity of the wave engine's Baum-Welch step (forward -> backward -> update without
materialising alpha/beta) against the CPU oracle's literal restatement of baumwelch.jl:25-309,
through the C ABI.  Bar from BASELINE.json: mu/sigma (and lp, pp) within 1e-6 relative; the
tests assert 1e-8 (observed differences are ~1e-11: exp/log implementations and summation order)."""
import numpy as np
import pytest

from conftest import four_templates, to_oracle_sm, two_templates

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def wave_engine(H):
    H.set_option("engine", H.ENGINE_WAVE)
    H.set_option("block", 0)
    H.set_option("halo", 0)
    yield
    H.set_option("engine", H.ENGINE_AUTO)
    H.set_option("block", 0)
    H.set_option("halo", 0)


def _compare_step(O, H, y, sm, mu, sigma, rtol=1e-8):
    osm = to_oracle_sm(O, sm)
    sm_n, mu_n, sig_n = H.train_step(y, sm, mu.copy(order="F"), sigma)
    osm_n, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), sigma)
    assert np.allclose(mu_n, omu, rtol=rtol, atol=1e-11), np.abs(mu_n - omu).max()
    assert abs(sig_n - osig) <= rtol * osig
    assert np.allclose(sm_n.transitions["lp"], osm_n.val, rtol=rtol, atol=1e-12)
    # pp = gamma[:,1] in the log domain; entries reach -1e3..-1e4, compare relatively
    assert np.allclose(sm_n.pi, opp, rtol=1e-8, atol=1e-8), np.abs(sm_n.pi - opp).max()
    return sm_n, mu_n, sig_n


@pytest.mark.parametrize("N,K,T,seed,block,halo", [
    (2, 30, 6_000, 1, 0, 0),
    (4, 60, 30_000, 2, 0, 0),
    (3, 60, 20_000, 3, 0, 0),          # README-size problem
    (4, 60, 40_001, 4, 512, 256),
    (1, 40, 9_000, 5, 0, 0),
    (8, 33, 12_000, 6, 0, 0),
    (16, 40, 10_000, 7, 0, 0),         # 16 rings (BASELINE config 5 ring count)
    (8, 128, 9_000, 8, 0, 0),          # BASELINE config 4 model shape (S = 1017)
    (10, 180, 9_000, 9, 0, 0),         # many long rings: statistics on the fp64 matrix cores
])
def test_em_step_matches_oracle(O, H, N, K, T, seed, block, halo):
    rng = np.random.default_rng(seed)
    H.set_option("block", block)
    H.set_option("halo", halo)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    pp = rng.uniform(1e-3, 4e-3, N) * min(1.0, 60.0 / K) * min(1.0, 4.0 / N)
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * rng.uniform(0.7, 1.2, N)[None, :])
    mu[0, :] = 0
    sm1, mu1, sig1 = _compare_step(O, H, y, sm, mu, 0.4)
    # a second step from the updated model (exercises set-up with the re-estimated lp)
    _compare_step(O, H, y, sm1, mu1, sig1)


def test_sixteen_long_rings_one_step(O, H):
    # 16 rings of 199 states (3185 states; BASELINE config 5 has 16 x 255): every column of the
    # 16-wide matrix-core statistics tile is a real ring.  One step only: the oracle needs ~10 s.
    N, K, T = 16, 200, 14_000
    rng = np.random.default_rng(10)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    pp = rng.uniform(2e-4, 3e-4, N)
    y = H.create_signal(T, 0.3, pp, temps, seed=10)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * rng.uniform(0.8, 1.1, N)[None, :])
    mu[0, :] = 0
    _compare_step(O, H, y, sm, mu, 0.4)


def test_random_initialisation_regime(O, H):
    # the reference's own start (baumwelch.jl:311-322): p0 = 2^(-3K/2), sigma = std(X), random
    # templates -> diffuse posteriors, tiny transition probabilities
    K, N, T = 40, 3, 12_000
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2)], 1))
    y = H.create_signal(T, 0.3, [0.004, 0.002], temps, seed=8)
    rng = np.random.default_rng(3)
    sigma = float(np.std(y, ddof=1))
    lp = np.log(np.full(N, 2.0 ** (-3 * K / 2)))
    sm = H.StateMatrix.create(N, K, lp, False)
    mu = np.ones((K, N), order="F")
    for i in range(N):
        mu[:, i] = H.create_spike_template(K, 3 * sigma * rng.random(),
                                           0.5 + 0.1 * rng.standard_normal(), 1.5 * rng.random())
    mu[0, :] = 0
    s, m, sg = sm, mu, sigma
    for _ in range(3):
        s, m, sg = _compare_step(O, H, y, s, m, sg, rtol=1e-7)


def test_plan_api_and_shard_additivity(O, H):
    # device-resident API: E-step statistics are plain sums, M-step runs from them on device
    import torch
    K, N, T = 60, 4, 60_000
    temps = four_templates(H, K)
    pp = [0.003, 0.001, 0.002, 0.0015]
    y = H.create_signal(T, 0.3, pp, temps, seed=12)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * 0.9)
    mu[0, :] = 0
    plan = H.Plan(T, sm, mu, 0.35)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    plan.mstep(stats, out, st)
    diag = plan.diagnostics(st)
    o = out.cpu().numpy()
    mu_n = o[:K * N].reshape((K, N), order="F")
    _, omu, osig, olp, _ = O.train_step(y, to_oracle_sm(O, sm), mu, 0.35)
    assert np.allclose(mu_n, omu, rtol=1e-8, atol=1e-11)
    assert abs(o[K * N] - osig) <= 1e-8 * osig
    assert np.allclose(o[K * N + 1:K * N + 1 + N], olp, rtol=1e-8)
    assert diag[3] == 0
    # sum_t sum_j gamma_t(j) = T  (every column of gamma is a distribution)
    NL = N * (K - 1)
    s = stats.cpu().numpy()
    assert abs(s[:NL].sum() + s[3 * NL + N] - T) < 1e-6 * T
    plan.close()


def test_em_driver_recovers_templates(O, H):
    # soft version of the reference's "Baum-Welch" test (test/runtests.jl:71-83) on the README
    # problem: from a perturbed start the EM loop (train_model driver, GPU steps) recovers both
    # templates within 1 % of their energy and the noise level
    K = 60
    temps = two_templates(H, K)
    pp = [0.003, 0.001]
    y = H.create_signal(30_000, 0.3, pp, temps, seed=1234)
    sm = H.StateMatrix.create(2, K, np.log([0.002, 0.002]), False)
    mu0 = np.asfortranarray(temps * np.array([0.6, 1.3])[None, :])
    mu0[0, :] = 0
    smn, mu, sig = H.train_model(y, sm, mu0, 0.5, 4)
    for i in range(2):
        assert np.sum((mu[:, i] - temps[:, i]) ** 2) / np.sum(temps[:, i] ** 2) < 0.01
    assert abs(sig - 0.3) < 0.01


def test_busy_signal_certificate_and_escalation(O, H):
    # nearly always-busy chain: the posterior-weighted boundary certificate (diag[3..6]) must
    # either pass with tiny errors or make the host-buffer entry point retry; either way the
    # returned step matches the oracle
    K, N, T = 60, 4, 40_000
    temps = four_templates(H, K)
    pp = [0.03, 0.02, 0.025, 0.02]
    y = H.create_signal(T, 0.3, pp, temps, seed=21)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * 0.9)
    mu[0, :] = 0
    H.set_option("block", 128)
    H.set_option("halo", 128)
    _compare_step(O, H, y, sm, mu, 0.35, rtol=1e-7)
    esc = H.get_option("last_escalations")
    import torch
    plan = H.Plan(T, sm, mu, 0.35)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    diag = plan.diagnostics(st)
    plan.close()
    print("busy signal: escalations", esc, "diag", diag)
    # the host-buffer call's first attempt builds this very plan (same options): it escalates exactly when a
    # forward/backward certificate fails, and a certificate fails exactly when its posterior-weighted error
    # (an L1 distance of two distributions, so at most 2) exceeds 1e-9
    assert (diag[3] + diag[5] > 0) == (esc > 0), (diag, esc)
    assert (max(diag[4], diag[6]) > 1e-9) == (diag[3] + diag[5] > 0), diag
    assert 0.0 <= max(diag[4], diag[6]) <= 2.0


def test_time_sharded_estep_equals_whole_recording(O, H):
    # one recording cut into 3 time shards (what 3 GPUs would hold): the shard statistics add up
    # to the statistics of the whole recording, and the M-step from the sum matches the oracle
    import torch
    K, N, T = 40, 3, 90_000
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2),
                                        H.create_spike_template(K, 2.5, 0.6, 0.25)], 1))
    pp = [0.004, 0.002, 0.003]
    y = H.create_signal(T, 0.3, pp, temps, seed=17)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * 0.9)
    mu[0, :] = 0
    st = torch.cuda.current_stream().cuda_stream
    whole = H.Plan(T, sm, mu, 0.35)
    dy = torch.from_numpy(y).cuda()
    ref = torch.zeros(whole.stats_len(), dtype=torch.float64, device="cuda")
    whole.estep(dy, ref, st)
    total = torch.zeros_like(ref)
    world = 3
    for rank in range(world):
        # shard edges carry a certificate: a certified chain boundary lies inside each halo
        plan, ys, (o_lo, o_hi) = H.dist.time_shard_plan(y, rank, world, sm, mu, 0.35, halo=256)
        assert plan.info()["block"] <= o_lo or rank == 0
        part = torch.zeros_like(ref)
        plan.estep(torch.from_numpy(ys).cuda(), part, st)
        d = plan.diagnostics(st)
        assert d[3] == 0 and d[5] == 0 and max(d[4], d[6]) < 1e-9, d
        total += part
        plan.close()
    # a shard whose halo is shorter than a chain has no certified boundary in front of its owned range: refused
    s_lo, s_hi, o_lo, o_hi, first, last = H.dist.time_shard(T, 1, world, halo=64)
    short = H.Plan(s_hi - s_lo, sm, mu, 0.35)
    with pytest.raises(H.HmmsortError):
        short.set_shard(o_lo, o_hi, first, last)
    short.close()
    torch.cuda.synchronize()
    r, t = ref.cpu().numpy(), total.cpu().numpy()
    assert np.allclose(t, r, rtol=1e-9, atol=1e-12), np.abs(t - r).max()
    out = torch.zeros(whole.mstep_len(), dtype=torch.float64, device="cuda")
    whole.mstep(total, out, st)
    o = out.cpu().numpy()
    _, omu, osig, olp, _ = O.train_step(y, to_oracle_sm(O, sm), mu, 0.35)
    assert np.allclose(o[:K * N].reshape((K, N), order="F"), omu, rtol=1e-8, atol=1e-11)
    assert abs(o[K * N] - osig) <= 1e-8 * osig and np.allclose(o[K * N + 1:K * N + 1 + N], olp, rtol=1e-8)
    whole.close()


def test_train_model_loop_stays_on_device(O, H):
    # baumwelch.jl:324-354 through the EM session (signal uploaded once, plan re-armed per step):
    # nsteps + nsteps//2 steps, callback(mu) before each of the first nsteps
    K, N, T = 30, 2, 20_000
    temps = two_templates(H, K)
    pp = [0.004, 0.003]
    y = H.create_signal(T, 0.3, pp, temps, seed=12)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu0 = np.asfortranarray(temps * 0.85)
    mu0[0, :] = 0
    seen = []
    sm_n, mu_n, sig_n = H.train_model(y, sm, mu0, 0.4, 2, lambda m: seen.append(m.copy()))
    osm, omu, osig = to_oracle_sm(O, sm), mu0.copy(order="F"), 0.4
    for _ in range(3):
        osm, omu, osig, _, _ = O.train_step(y, osm, omu, osig)
    assert len(seen) == 2 and np.array_equal(seen[0], mu0)
    assert np.allclose(mu_n, omu, rtol=1e-8, atol=1e-11) and abs(sig_n - osig) <= 1e-8 * osig
    assert np.allclose(sm_n.transitions["lp"], osm.val, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("N,K,T", [(4, 60, 300_000), (3, 17, 50_001), (4, 65, 120_000),
                                   (8, 128, 200_000), (5, 50, 90_001), (8, 65, 100_000), (6, 100, 120_000), (7, 129, 150_000)])
def test_fused_statistics_equal_the_separate_kernel(H, N, K, T, monkeypatch):
    # 3-4 rings of at most 64 states, 5-8 rings of at most 128 (BASELINE config 4's shape: two / four accumulator
    # tiles of 32 lags): the backward sweep accumulates the spike-triggered sums itself (matrix cores, LDS rings);
    # HMMSORT_GSUM_SEPARATE forces the stand-alone statistics kernel on the same posteriors
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 2.5 + 0.3 * i, 0.3 + 0.08 * i, 0.2) for i in range(N)], 1))
    pp = ([0.003, 0.001, 0.002, 0.0015] * 2)[:N]
    pp = [p * min(1.0, 60.0 / K) for p in pp]
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    y = H.create_signal(T, 0.3, pp, temps, seed=N * K)
    mu0 = np.asfortranarray(temps * 0.9)
    mu0[0, :] = 0
    H.set_option("plan_cache", 0)
    try:
        fused = H.train_step(y, sm, mu0.copy(order="F"), 0.35)
        monkeypatch.setenv("HMMSORT_GSUM_SEPARATE", "1")
        separate = H.train_step(y, sm, mu0.copy(order="F"), 0.35)
    finally:
        monkeypatch.delenv("HMMSORT_GSUM_SEPARATE", raising=False)
        H.set_option("plan_cache", 4)
    assert np.allclose(fused[1], separate[1], rtol=1e-11, atol=1e-13), np.abs(fused[1] - separate[1]).max()
    assert abs(fused[2] - separate[2]) <= 1e-12 * separate[2]
    assert np.allclose(fused[0].transitions["lp"], separate[0].transitions["lp"], rtol=1e-12, atol=1e-14)
