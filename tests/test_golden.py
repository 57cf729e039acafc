"""Oracle regression guard against the committed fixtures (tests/golden/*.npz, produced by
tests/golden/make_golden.py from the oracle itself).  CPU only."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_case(O, path):
    g = np.load(path)
    sm = O.StateMatrix(np.asfortranarray(g["states"]), g["src"], g["dst"], g["val"],
                       np.zeros(g["states"].shape[1]), int(g["K"]), int(g["N"]),
                       g["states"].shape[1], bool(g["ov"]))
    return g, sm


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_fixture(O, path):
    g, sm = load_case(O, path)
    temps, y = np.asfortranarray(g["temps"]), g["y"]
    # the transition list itself
    ref = O.state_matrix(int(g["N"]), int(g["K"]), np.log(g["pp"]), bool(g["ov"]))
    assert np.array_equal(ref.src, g["src"]) and np.array_equal(ref.val, g["val"])
    x, ll = O.viterbi(y, sm, temps, 0.3)
    assert np.array_equal(x, g["x"]) and ll == float(g["ll"])
    a = O.forward(y, sm, temps, 0.3)[:, g["ab_cols"]]
    b = O.backward(y, sm, temps, 0.3)[:, g["ab_cols"]]
    assert np.allclose(a, g["alpha"], rtol=1e-13, atol=0)
    assert np.allclose(b, g["beta"], rtol=1e-13, atol=1e-300)
    mu = np.asfortranarray(temps * 0.85)
    mu[0, :] = 0
    sig, smi = 0.4, sm
    for step in (1, 2, 3):
        smi, mu, sig, lp, pp = O.train_step(y, smi, mu, sig)
        if step in (1, 3):
            assert np.allclose(mu, g["em%d_mu" % step], rtol=1e-12, atol=1e-14)
            assert np.isclose(sig, float(g["em%d_sigma" % step]), rtol=1e-13)
            assert np.allclose(lp, g["em%d_lp" % step], rtol=1e-12)
