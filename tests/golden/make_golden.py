"""Generates the golden fixtures in this directory FROM THE CPU ORACLE (oracle/hmm_oracle.c).

The reference (Julia) cannot run in the build image and holds no golden alpha/beta/path vectors
of its own, so these fixtures pin the oracle's current behaviour (regression guard) and give the
GPU tests fixed inputs/outputs; they are NOT outputs of the reference ("parity unpinned" for
these numerics).  Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
import hmmsort_amd as H  # noqa: E402  (synthetic generator + host-side state space only)


def case(name, N, K, T, ov, seed, pp, amps, full_ab):
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = O.state_matrix(N, K, np.log(pp), ov)
    x, ll = O.viterbi(y, sm, temps, 0.3)
    a = O.forward(y, sm, temps, 0.3)
    b = O.backward(y, sm, temps, 0.3)
    out = dict(N=N, K=K, T=T, ov=int(ov), seed=seed, pp=np.array(pp), temps=temps, y=y,
               src=sm.src, dst=sm.dst, val=sm.val, states=sm.states, x=x, ll=ll)
    cols = np.arange(T) if full_ab else np.unique(np.r_[np.arange(0, T, 97), T - 1])
    out.update(ab_cols=cols, alpha=a[:, cols], beta=b[:, cols])
    # EM: 1 and 3 steps from a perturbed start (mu row 1 forced to 0, baumwelch.jl:320)
    mu = np.asfortranarray(temps * 0.85)
    mu[0, :] = 0
    sig = 0.4
    smi = sm
    for step in (1, 2, 3):
        smi, mu, sig, lp, ppv = O.train_step(y, smi, mu, sig)
        if step in (1, 3):
            out["em%d_mu" % step] = mu.copy()
            out["em%d_sigma" % step] = sig
            out["em%d_lp" % step] = lp
            out["em%d_pp" % step] = ppv
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "S=%d R=%d ll=%r" % (sm.nstates, len(sm.src), ll))


if __name__ == "__main__":
    two = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2)]
    case("n2k5_ov", 2, 5, 200, True, 11, [0.02, 0.01], two, True)
    case("n2k5_noov", 2, 5, 200, False, 12, [0.02, 0.01], two, True)
    case("n3k60", 3, 60, 2000, False, 13, [0.003, 0.001, 0.002], two + [(2.5, 0.6, 0.25)], False)
    case("n2k20", 2, 20, 1500, False, 14, [0.01, 0.004], two, False)
    # long enough for the time-parallel blocked engine (overlap model, T >= 4096)
    case("n2k16_ov", 2, 16, 6000, True, 15, [0.01, 0.006], two, False)
