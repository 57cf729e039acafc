"""Ragged and boundary-sized inputs for the wave engine (chain/tile boundaries, shortest signals
and rings it accepts, non-finite samples), against the CPU oracle through the C ABI."""
import numpy as np
import pytest

from conftest import to_oracle_sm, two_templates

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


# lengths around multiples of the chain length (B = 256 for these ring lengths), the minimum the
# engine takes (512), one chain, last chain of exactly L samples, ...
LENGTHS = [512, 513, 767, 768, 769, 1023, 1024, 1025, 256 * 7 + 19, 256 * 7 + 20, 256 * 7 + 255,
           4096, 5000]


@pytest.mark.parametrize("T", LENGTHS)
def test_ragged_lengths(O, H, T):
    K, N = 21, 2   # ring length 20
    temps = two_templates(H, K)
    pp = [0.01, 0.006]
    y = H.create_signal(T, 0.3, pp, temps, seed=T)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    osm = to_oracle_sm(O, sm)
    x, ll = H.viterbi(y, sm, temps, 0.3)
    xo, llo = O.viterbi(y, osm, temps, 0.3)
    assert np.array_equal(x, xo) and abs(ll - llo) <= 1e-9 * abs(llo)
    mu = np.asfortranarray(temps * 0.85)
    mu[0, :] = 0
    smn, mun, sgn = H.train_step(y, sm, mu.copy(order="F"), 0.4)
    osmn, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), 0.4)
    assert np.allclose(mun, omu, rtol=1e-8, atol=1e-11) and abs(sgn - osig) <= 1e-8 * osig
    assert np.allclose(smn.transitions["lp"], osmn.val, rtol=1e-8)
    assert np.allclose(smn.pi, opp, rtol=1e-8, atol=1e-8)


def test_signal_ending_and_starting_inside_a_spike(O, H):
    # the first and last samples sit in the middle of a template: exercises the reference's
    # emission-only first column and beta = 0 terminal condition (virtual onsets / truncated rings)
    K, N, T = 40, 2, 3000
    temps = two_templates(H, K)
    pp = [0.004, 0.003]
    y = H.create_signal(T, 0.3, pp, temps, seed=2)
    y[:25] += temps[15:40, 0]        # tail of a spike that started before the recording
    y[-20:] += temps[:20, 1]         # head of a spike that runs past the end
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    osm = to_oracle_sm(O, sm)
    x, ll = H.viterbi(y, sm, temps, 0.3)
    xo, llo = O.viterbi(y, osm, temps, 0.3)
    assert np.array_equal(x, xo) and abs(ll - llo) <= 1e-9 * abs(llo)
    assert x[0] > 1 and x[-1] > 1    # the decode really starts and ends inside rings
    mu = np.asfortranarray(temps * 0.9)
    mu[0, :] = 0
    smn, mun, sgn = H.train_step(y, sm, mu.copy(order="F"), 0.35)
    osmn, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), 0.35)
    assert np.allclose(mun, omu, rtol=1e-8, atol=1e-11) and abs(sgn - osig) <= 1e-8 * osig
    assert np.allclose(smn.pi, opp, rtol=1e-8, atol=1e-8)


def test_non_finite_samples_do_not_hang(H):
    # garbage in, garbage out -- but every kernel terminates and the call returns
    K, N, T = 30, 2, 20_000
    temps = two_templates(H, K)
    pp = [0.004, 0.003]
    y = H.create_signal(T, 0.3, pp, temps, seed=3)
    y[5000] = np.nan
    y[12000] = np.inf
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    H.set_option("escalate", 0)
    try:
        x, ll = H.viterbi(y, sm, temps, 0.3)
        assert x.shape == (T,) and x.min() >= 1 and x.max() <= sm.nstates
        mu = np.asfortranarray(temps.copy())
        H.train_step(y, sm, mu, 0.3)
    finally:
        H.set_option("escalate", 1)


def test_large_amplitude_and_tiny_sigma(O, H):
    # posteriors that underflow almost everywhere, ring scores of several thousand nats
    K, N, T = 30, 2, 8000
    temps = two_templates(H, K) * 4.0
    pp = [0.004, 0.003]
    y = H.create_signal(T, 0.05, pp, temps, seed=4)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    osm = to_oracle_sm(O, sm)
    x, ll = H.viterbi(y, sm, temps, 0.05)
    xo, llo = O.viterbi(y, osm, temps, 0.05)
    assert np.array_equal(x, xo)
    mu = np.asfortranarray(temps * 0.97)
    mu[0, :] = 0
    smn, mun, sgn = H.train_step(y, sm, mu.copy(order="F"), 0.08)
    osmn, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), 0.08)
    assert np.allclose(mun, omu, rtol=1e-7, atol=1e-10) and abs(sgn - osig) <= 1e-7 * osig


def test_dpp_scans_match_shuffle_references(H):
    """the cross-lane primitives of the wave engine (DPP row_shr / row_bcast / wave_shr moves) against
    their shuffle-based references and against numpy"""
    import ctypes as C
    rng = np.random.default_rng(5)
    inp = np.concatenate([rng.normal(-1, 2, 64), rng.normal(0, 5, 64), rng.uniform(0.2, 1.0, 64),
                          rng.uniform(0, 2, 64)])
    inp[64 + 7] = -np.inf
    out = np.zeros(896)
    fn = H._lib.lib().hmmsort_selftest_scans
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    assert fn(inp.ctypes.data, out.ctypes.data) == 0
    a, b, a2, b2, c, d, c2, d2 = [out[64 * i:64 * i + 64] for i in range(8)]
    # same function, different association order of the 64-lane scan (rows of 16 first): equal to rounding
    for u, v in ((a, a2), (b, b2), (c, c2), (d, d2)):
        assert np.allclose(u, v, rtol=1e-13, atol=1e-13)
    # sequential semantics: f_j(x) = max(x + a_j, b_j) composed over j = 0..i
    x = -3.25
    for i in range(64):
        x = max(x + inp[i], inp[64 + i])
        assert abs(max(-3.25 + a[i], b[i]) - x) <= 1e-12 * max(1.0, abs(x))
    x = 0.75
    for i in range(64):
        x = inp[128 + i] * x + inp[192 + i]
        assert abs((c[i] * 0.75 + d[i]) - x) <= 1e-12 * max(1.0, abs(x))
    assert np.array_equal(out[512:576], out[576:640])
    assert out[512] == 123.5 and np.array_equal(out[513:576], inp[:63])
    assert np.all(out[640:704] == inp[63])
    # single-precision envelope scan (forward/backward scale) and float lane shift
    assert np.allclose(out[704:768], a2, rtol=1e-5, atol=1e-4) and np.allclose(out[768:832][np.isfinite(b2)], b2[np.isfinite(b2)], rtol=1e-5, atol=1e-4)
    assert out[832] == 7.25 and np.array_equal(out[833:896], inp[:63].astype(np.float32).astype(np.float64))
