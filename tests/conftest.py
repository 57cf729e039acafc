import os
import sys

import numpy as np
import pytest

os.environ.setdefault("HMMSORT_POISON", "1")  # NaN-fill GPU work arrays (catches unwritten reads)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def H():
    """The product package (libhmmsort_hip.so behind it)."""
    import hmmsort_amd
    return hmmsort_amd


def two_templates(H, K=60):
    """The two templates every reference test uses (test/runtests.jl:19-21)."""
    t1 = H.create_spike_template(K, 3.0, 0.8, 0.2)
    t2 = H.create_spike_template(K, 4.0, 0.3, 0.2)
    return np.asfortranarray(np.stack([t1, t2], axis=1))


def four_templates(H, K=60):
    """BASELINE config 2/3 templates."""
    p = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    return np.asfortranarray(np.stack([H.create_spike_template(K, *q) for q in p], axis=1))


def to_oracle_sm(O, sm):
    """product StateMatrix -> oracle StateMatrix (same arrays, split transition columns)."""
    tr = sm.transitions
    return O.StateMatrix(np.asfortranarray(sm.states), np.ascontiguousarray(tr["src"]),
                         np.ascontiguousarray(tr["dst"]), np.ascontiguousarray(tr["lp"]),
                         sm.pi, sm.K, sm.N, sm.nstates, sm.resolve_overlaps)
