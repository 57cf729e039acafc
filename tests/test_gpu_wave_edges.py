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


def _per_source_list(H, sm, N, K, rng):
    """A ring-model list whose exit->entry log-probabilities (b,L) -> (a,1) depend on the SOURCE ring b: the
    wave engine's `uniform cx` shortcut (one top-3 over the ring exits serves every junction) does not apply
    and the kernels take their O(N^2) junction code (kw_vit/kw_fwd/kw_bwd with UC = false).  Lists built by
    types.jl:94-113 reach that code only when the neuron-order sums differ in the last bit between sources."""
    L = K - 1
    tr = sm.transitions.copy()
    n = 0
    for i in range(len(tr)):
        s, d = int(tr["src"][i]), int(tr["dst"][i])
        if s > 1 and d > 1 and (s - 2) % L == L - 1 and (d - 2) % L == 0:
            tr["lp"][i] += rng.uniform(-0.7, 0.7)
            n += 1
    assert n == N * (N - 1)
    return H.StateMatrix(sm.states, tr, sm.pi, sm.K, sm.N, sm.nstates, False)


@pytest.mark.parametrize("N,K,T,seed", [(4, 60, 150_000, 1), (3, 30, 40_000, 2), (8, 40, 60_000, 3), (8, 24, 30_000, 5),
                                        (6, 20, 30_000, 6), (12, 24, 50_000, 4), (13, 24, 30_000, 7)])
def test_exit_to_entry_values_that_depend_on_the_source_ring(O, H, N, K, T, seed):
    rng = np.random.default_rng(seed)
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, base[i % 4][0] * (1 + 0.15 * (i // 4)),
                                                                 base[i % 4][1] + 0.04 * (i // 4), base[i % 4][2])
                                        for i in range(N)], 1))
    # busy enough that spikes follow each other directly (the exit -> entry transitions are on the paths)
    pp = rng.uniform(0.004, 0.012, N) * min(1.0, 4.0 / N)
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    # the generator never lets spikes abut (it races only while silent): add pairs that do, so that the decoded
    # paths take exit -> entry transitions (a,L) -> (b,1)
    L = K - 1
    for i in range(12):
        t0 = 2000 + i * (T - 4000) // 12
        a, b = i % N, (i + 1 + i // N) % N
        if a == b:
            b = (b + 1) % N
        y[t0:t0 + L] += 1.5 * temps[1:, a]
        y[t0 + L:t0 + 2 * L] += 1.5 * temps[1:, b]
    sm = _per_source_list(H, H.StateMatrix.create(N, K, np.log(pp), False), N, K, rng)
    osm = to_oracle_sm(O, sm)
    if N > 8:
        # the wave engine's per-source junction code is built for up to 8 rings: such a list is refused by the
        # wave engine and AUTO hands it to the blocked engine (any transition list)
        with pytest.raises(H.HmmsortError):
            H.viterbi(y, sm, temps, 0.3)
        H.set_option("engine", H.ENGINE_AUTO)
    x, ll = H.viterbi(y, sm, temps, 0.3)
    xo, llo = O.viterbi(y, osm, temps, 0.3)
    # the paths do use ring -> ring transitions
    direct = np.count_nonzero((xo[:-1] > 1) & ((xo[:-1] - 2) % L == L - 1) & (xo[1:] > 1))
    assert direct > 3, direct
    assert np.array_equal(x, xo), int(np.count_nonzero(x != xo))
    assert abs(ll - llo) <= 1e-9 * abs(llo)
    mu = np.asfortranarray(temps * rng.uniform(0.85, 1.1, N)[None, :])
    mu[0, :] = 0
    smn, mun, sgn = H.train_step(y, sm, mu.copy(order="F"), 0.4)
    osmn, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), 0.4)
    assert H.get_option("last_escalations") == 0
    assert np.allclose(mun, omu, rtol=1e-8, atol=1e-11), np.abs(mun - omu).max()
    assert abs(sgn - osig) <= 1e-8 * osig
    assert np.allclose(smn.transitions["lp"], osmn.val, rtol=1e-8)
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
