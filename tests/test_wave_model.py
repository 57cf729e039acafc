"""The numpy specification of the wave engine (tests/wave_model.py) against the CPU oracle: Viterbi
path bit-identical, forward values and the Baum-Welch step within 1e-9 -- with one chain (the plain
recursions in the scaled representation) and with several chains (warm-up, per-chain normaliser,
exact hand-off on a failed certificate)."""
import numpy as np
import pytest

import wave_model as WM
from conftest import to_oracle_sm


def _case(H, N, K, T, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *base[i % 4]) for i in range(N)], 1))
    temps *= (1 + 0.1 * np.arange(N))[None, :]
    pp = rng.uniform(2e-3, 6e-3, N) * scale
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    return y, sm, temps


@pytest.mark.parametrize("N,K,T,B,Hh,seed", [
    (2, 20, 1500, 4096, 0, 1),      # one chain
    (3, 20, 2400, 512, 128, 2),     # several chains, L < 64
    (2, 80, 3000, 1024, 384, 3),    # L > 64: super-steps of 64 lanes
])
def test_viterbi_model_matches_oracle(O, H, N, K, T, B, Hh, seed):
    y, sm, temps = _case(H, N, K, T, seed)
    osm = to_oracle_sm(O, sm)
    xo, _ = O.viterbi(y, osm, temps, 0.3)
    m = WM.Ring(sm, temps, 0.3)
    x, spreads, nflag = WM.vit_decode(y, m, B, Hh, thr=1e-9)
    assert np.array_equal(x, xo)
    assert nflag == 0


def test_viterbi_model_short_warmup_is_repaired(O, H):
    y, sm, temps = _case(H, 2, 20, 2000, 5, scale=4.0)
    osm = to_oracle_sm(O, sm)
    xo, _ = O.viterbi(y, osm, temps, 0.3)
    m = WM.Ring(sm, temps, 0.3)
    x, spreads, _ = WM.vit_decode(y, m, 256, 8, thr=0.0)   # 8-sample warm-up: certificates fail
    assert max(spreads) > 1e-6
    assert np.array_equal(x, xo)


@pytest.mark.parametrize("N,K,T,B,Hh,seed", [
    (2, 20, 1200, 4096, 0, 11),
    (3, 20, 2000, 512, 192, 12),
    (2, 80, 2400, 1024, 512, 13),
])
def test_estep_model_matches_oracle(O, H, N, K, T, B, Hh, seed):
    y, sm, temps = _case(H, N, K, T, seed)
    osm = to_oracle_sm(O, sm)
    mu0 = np.asfortranarray(temps * 0.9)
    mu0[0, :] = 0
    m = WM.Ring(sm, mu0, 0.4)
    mu, sigma, lp_new, pp, g0, rho = WM.estep(y, m, B, Hh)
    _, omu, osig, olp, opp = O.train_step(y, osm, mu0.copy(order="F"), 0.4)
    assert np.allclose(mu, omu, rtol=1e-9, atol=1e-12), np.abs(mu - omu).max()
    assert abs(sigma - osig) <= 1e-10 * osig
    assert np.allclose(lp_new, olp, rtol=1e-9, atol=1e-12)
    assert np.allclose(pp, opp, rtol=1e-9, atol=1e-9)
    # posterior mass: silent + every ring state = 1 at every sample
    L = K - 1
    mass = g0.copy()
    cs = np.concatenate([np.zeros((N, 1)), np.cumsum(rho, 1)], 1)
    for t in range(L, T):
        mass[t] += (cs[:, t + 1] - cs[:, t + 1 - L]).sum()
    assert np.allclose(mass[L:], 1.0, rtol=0, atol=1e-9)


def test_forward_model_matches_oracle_alpha(O, H):
    y, sm, temps = _case(H, 2, 20, 800, 21)
    osm = to_oracle_sm(O, sm)
    m = WM.Ring(sm, temps, 0.3)
    Rf, V = WM.ring_scores(y, m)
    la0, fv, fref = WM.fwd_chain(y, Rf, V, m, len(y), 4096, 0, 0)
    alpha = O.forward(y, osm, temps, 0.3)
    t = np.arange(len(y))
    got = np.array([la0[i] for i in t]) + m.A * (t + 1)
    assert np.allclose(got, alpha[0], rtol=1e-11, atol=1e-9)
    # first state of ring a at time t: alpha = lp_a(t) - (R_a(t) - q(y_t; mean(a,1)))
    for a in range(2):
        for tt in (5, 100, 333):
            lpa = fref[tt] + m.sc[a] + np.log(fv[tt][a])
            d = y[tt] - m.mean[a, 0]
            want = alpha[1 + a * m.L, tt] - m.A * (tt + 1)
            assert abs((lpa - (d * d) / m.den) - want) <= 1e-9 * max(1.0, abs(want))


@pytest.mark.parametrize("lp_dead", [None, -400.0, -900.0])
def test_estep_model_random_start_and_vanishing_template(O, H, lp_dead):
    """the reference's random start (p0 = 2^(-3K/2), baumwelch.jl:311-322) and a template whose entry
    probability is far below the double range: lp stays finite as in the reference (its log-domain
    folds, baumwelch.jl:254-261), the other templates are unaffected"""
    N, K, T = 3, 20, 1500
    y, sm, temps = _case(H, N, K, T, 31)
    lp = np.full(N, np.log(2.0 ** (-3 * K / 2)))
    if lp_dead is not None:
        lp[1] = lp_dead
    sm = H.StateMatrix.create(N, K, lp, False)
    osm = to_oracle_sm(O, sm)
    rng = np.random.default_rng(3)
    mu0 = np.asfortranarray(temps * rng.uniform(0.6, 1.3, N)[None, :])
    mu0[0, :] = 0
    m = WM.Ring(sm, mu0, 0.45)
    mu, sigma, lp_new, pp, g0, rho = WM.estep(y, m, 512, 160)
    _, omu, osig, olp, opp = O.train_step(y, osm, mu0.copy(order="F"), 0.45)
    assert np.all(np.isfinite(olp)) and np.all(np.isfinite(lp_new))
    assert np.allclose(lp_new, olp, rtol=1e-9, atol=1e-9), (lp_new, olp)
    live = [a for a in range(N) if lp_dead is None or a != 1 or lp_dead > -600]
    assert np.allclose(mu[:, live], omu[:, live], rtol=1e-8, atol=1e-11)
    if len(live) == N:
        assert abs(sigma - osig) <= 1e-9 * osig


def test_estep_model_chain_starting_inside_a_large_spike(O, H):
    """a warm-up that starts in the middle of a spike: the scale of the silent path falls by
    thousands of nats while the delay line still holds its empty entries (mantissa 0, scale 0):
    0 * exp(+big) must not turn into NaN (found on the GPU: 16 x 200 model, T = 14 000)"""
    N, K, T, B, Hh = 2, 80, 3000, 1024, 512
    y, sm, temps = _case(H, N, K, T, 41)
    for c in (1, 2):
        t0 = c * B - (1 + ((Hh + 63) // 64) * 64) - 20
        y[t0:t0 + K] += 2.5 * temps[:, 1]
    osm = to_oracle_sm(O, sm)
    mu0 = np.asfortranarray(temps * 0.9)
    mu0[0, :] = 0
    m = WM.Ring(sm, mu0, 0.25)
    mu, sigma, lp_new, pp, g0, rho = WM.estep(y, m, B, Hh)
    _, omu, osig, olp, opp = O.train_step(y, osm, mu0.copy(order="F"), 0.25)
    assert np.all(np.isfinite(mu[1:]))
    # the warm-up really starts in the wrong state here: its error has decayed to ~1e-9, not to rounding
    assert np.allclose(mu, omu, rtol=1e-7, atol=1e-10) and abs(sigma - osig) <= 1e-8 * osig
    assert np.allclose(lp_new, olp, rtol=1e-7, atol=1e-10)
