"""Randomised parity sweep (scripts/fuzz_gpu.py): random model shapes (1-6 templates, 2-69 states,
with and without overlap states), firing rates, noise levels and signal lengths (300-40 000
samples), decode + one EM step against the oracle.  Seed 2 contains the case (N=4, K=20,
T=32 004) that exposed the first-sample tie between 'ring in its last phase' states, now decided
with the reference's own operations (k_first_state)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("seed", [2, 3])
def test_random_models_match_the_oracle(O, H, seed):
    import fuzz_gpu
    failures = fuzz_gpu.run(150, seed, verbose=False)
    assert not failures, failures
