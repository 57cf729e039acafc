"""Time-parallel E-step for overlap models (generic_estep.hip): forward / backward / update of
baumwelch.jl:25-98, 205-309 over the pair-state space of types.jl:78-90, run block-parallel with a
certified warm-up and without S x T arrays.  Checked against the CPU oracle (small cases), against the strict
engine (the reference's own op order on materialised alpha/beta) at 10^6 samples, and through the plan API."""
import numpy as np
import pytest

from conftest import to_oracle_sm, two_templates

pytestmark = pytest.mark.gpu


def overlap_case(H, N, K, T, seed):
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *base[i]) for i in range(N)], 1))
    pp = [0.012, 0.008, 0.006][:N]
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), True)
    mu0 = np.asfortranarray(temps * 0.85)
    mu0[0, :] = 0
    return y, sm, mu0


def step(H, engine, y, sm, mu0, sigma):
    H.set_option("engine", engine)
    try:
        out = H.train_step(y, sm, mu0.copy(order="F"), sigma)
        return out, H.get_option("last_escalations")
    finally:
        H.set_option("engine", H.ENGINE_AUTO)


@pytest.mark.parametrize("N,K,T", [(2, 20, 20_000), (3, 12, 12_000), (2, 33, 9_000)])
def test_blocked_estep_matches_oracle(O, H, N, K, T):
    y, sm, mu0 = overlap_case(H, N, K, T, seed=N * 10 + K)
    (sm_n, mu_n, sig_n), esc = step(H, H.ENGINE_BLOCKED, y, sm, mu0, 0.4)
    assert esc == 0
    osm_n, omu, osig, olp, opp = O.train_step(y, to_oracle_sm(O, sm), mu0.copy(order="F"), 0.4)
    assert np.allclose(mu_n, omu, rtol=1e-8, atol=1e-11), np.abs(mu_n - omu).max()
    assert abs(sig_n - osig) <= 1e-9 * osig
    assert np.allclose(sm_n.transitions["lp"], osm_n.val, rtol=1e-8, atol=1e-11)
    # AUTO picks the blocked engine for an overlap model of this length
    (sm_a, mu_a, sig_a), _ = step(H, H.ENGINE_AUTO, y, sm, mu0, 0.4)
    assert np.array_equal(mu_a, mu_n) and sig_a == sig_n


def test_blocked_estep_vs_strict_engine_at_1e6_samples(H):
    # N=2, K=60 with overlaps (3600 states), 10^6 samples.  The strict engine
    # materialises alpha, beta, gamma: 3 x 28.8 GB
    y, sm, mu0 = overlap_case(H, 2, 60, 1_000_000, seed=3)
    assert sm.nstates == 3600
    import time
    t0 = time.perf_counter()
    (sm_b, mu_b, sig_b), esc = step(H, H.ENGINE_BLOCKED, y, sm, mu0, 0.4)
    t1 = time.perf_counter()
    (sm_s, mu_s, sig_s), _ = step(H, H.ENGINE_STRICT, y, sm, mu0, 0.4)
    t2 = time.perf_counter()
    print("overlap E-step, 3600 states x 1e6 samples: blocked %.3f s (%d escalations), strict %.1f s"
          % (t1 - t0, esc, t2 - t1))
    assert esc == 0
    assert np.allclose(mu_b, mu_s, rtol=1e-8, atol=1e-11), np.abs(mu_b - mu_s).max()
    assert abs(sig_b - sig_s) <= 1e-9 * sig_s
    assert np.allclose(sm_b.transitions["lp"], sm_s.transitions["lp"], rtol=1e-8, atol=1e-11)
    H.shutdown()


def test_blocked_plan_estep_mstep_and_short_warmup_escalates(O, H):
    import torch
    y, sm, mu0 = overlap_case(H, 2, 20, 30_000, seed=9)
    osm_n, omu, osig, olp, opp = O.train_step(y, to_oracle_sm(O, sm), mu0.copy(order="F"), 0.4)
    H.set_option("engine", H.ENGINE_BLOCKED)
    try:
        plan = H.Plan(len(y), sm, mu0, 0.4)
        assert plan.info()["engine"] == H.ENGINE_BLOCKED
        dy = torch.from_numpy(y).cuda()
        stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
        out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
        plan.estep(dy, stats)
        plan.mstep(stats, out)
        dg = plan.diagnostics()
        assert dg[3] == 0 and dg[5] == 0 and max(dg[4], dg[6]) < 1e-9, dg
        o = out.cpu().numpy()
        KN = sm.K * sm.N
        assert np.allclose(o[:KN].reshape(sm.N, sm.K).T, omu, rtol=1e-8, atol=1e-11)
        assert abs(o[KN] - osig) <= 1e-9 * osig
        s = stats.cpu().numpy()
        assert abs(s[:sm.nstates].sum() - len(y)) < 1e-7 * len(y)        # posterior mass = T
        plan.close()
        # a warm-up of 64 samples (the smallest the geometry allows) is one ring length at K = 60 and cannot
        # forget its start: the certificates say so and em_step widens it
        y2, sm2, mu2 = overlap_case(H, 2, 60, 20_000, seed=11)
        (sm_s, mu_s, sig_s), _ = step(H, H.ENGINE_STRICT, y2, sm2, mu2, 0.4)
        H.set_option("engine", H.ENGINE_BLOCKED)
        H.set_option("halo", 8)
        (sm_n, mu_n, sig_n) = H.train_step(y2, sm2, mu2.copy(order="F"), 0.4)
        assert H.get_option("last_escalations") >= 1
        assert np.allclose(mu_n, mu_s, rtol=1e-8, atol=1e-11) and abs(sig_n - sig_s) <= 1e-9 * sig_s
    finally:
        H.set_option("halo", 0)
        H.set_option("engine", H.ENGINE_AUTO)
        H.shutdown()


def test_blocks_shorter_than_the_warmup(O, H):
    # a requested block of 128 samples under the default warm-up of 256: the warm-up of the first blocks
    # reaches the start of the data (a fuzz case: the sweep started before sample 0)
    y, sm, mu0 = overlap_case(H, 2, 12, 6_000, seed=21)
    osm_n, omu, osig, olp, opp = O.train_step(y, to_oracle_sm(O, sm), mu0.copy(order="F"), 0.4)
    H.set_option("block", 128)
    try:
        (sm_n, mu_n, sig_n), esc = step(H, H.ENGINE_BLOCKED, y, sm, mu0, 0.4)
    finally:
        H.set_option("block", 0)
        H.shutdown()
    assert np.allclose(mu_n, omu, rtol=1e-8, atol=1e-11) and abs(sig_n - osig) <= 1e-9 * osig
