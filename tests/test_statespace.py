"""Host-side state space of the product (closed-form, O(R)) against the oracle's literal
restatement of types.jl:65-127 (all-pairs scan).  Bit-exact: same list, same order, same doubles.
CPU only (these C-ABI functions need no GPU)."""
import numpy as np
import pytest


CASES = [(1, 2, False), (1, 5, True), (2, 5, True), (2, 5, False), (3, 4, True), (3, 60, False),
         (4, 60, False), (2, 60, True), (5, 3, True), (8, 128, False), (4, 9, True)]


@pytest.mark.parametrize("N,K,ov", CASES)
def test_transitions_bit_exact(O, H, N, K, ov):
    rng = np.random.default_rng(N * 1000 + K)
    lp = np.log(rng.uniform(1e-4, 2e-2, N))
    ref = O.state_matrix(N, K, lp, ov)
    got = H.StateMatrix.create(N, K, lp, ov)
    assert got.nstates == ref.nstates
    assert np.array_equal(got.states, ref.states)
    assert np.array_equal(got.transitions["src"], ref.src)
    assert np.array_equal(got.transitions["dst"], ref.dst)
    assert np.array_equal(got.transitions["lp"], ref.val)  # bitwise


def test_lp_longer_than_N_quirk(O, H):
    # with overlaps update() returns xb[2:end] longer than N (baumwelch.jl:226,265); the
    # constructor then uses lp[1:N] per neuron but sum(lp) over ALL entries (types.jl:96)
    N, K = 2, 4
    lp = np.log([0.01, 0.02, 1e-4])
    ref = O.state_matrix_from_states(O.generate_states(N, K, True), np.zeros(1), K, lp, True)
    got = H.StateMatrix.from_states(ref.states, np.zeros(1), K, lp, True)
    assert np.array_equal(got.transitions["lp"], ref.val)
    assert np.array_equal(got.transitions["dst"], ref.dst)


def test_bad_arguments(H):
    with pytest.raises(H.HmmsortError):
        H.StateMatrix.create(0, 5, np.array([]), False)
    with pytest.raises(H.HmmsortError):
        H.StateMatrix.create(3, 200, np.log([0.1, 0.1, 0.1]), True)  # > 32767 states (Int16 ids)
