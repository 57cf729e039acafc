"""Synthetic spike-train signals in the shape the reference's tests and README use.

Restates `create_spike_template` (reference src/utils.jl:51-55) and the generative rule of
`create_signal` (src/utils.jl:57-86).  Julia's MersenneTwister/randn stream cannot be reproduced
without Julia, so the random stream is this build's own (numpy PCG64, seed recorded by callers);
the *rule* is the reference's: Gaussian noise, one neuron active at a time, a per-sample
Bernoulli race in neuron order while silent, template emitted from its first row in the sample
the neuron fires.
"""
import numpy as np


def create_spike_template(nstates, a=1.0, b=0.8, c=0.2):
    """utils.jl:51-55:  x = range(0, stop=1.5, length=nstates); a*sin(2*pi*x)*exp(-(b-x)^2/c)."""
    i = np.arange(nstates, dtype=np.float64)
    x = (i * 1.5) / (nstates - 1)
    return (a * np.sin((2 * np.pi) * x)) * np.exp(-((b - x) ** 2) / c)


def create_signal(n, sigma, pp, templates, seed=1234, return_states=False):
    """utils.jl:57-86.  templates: (nstates, ncells).  Returns S (and the true onset list)."""
    templates = np.asarray(templates, dtype=np.float64)
    nstates, ncells = templates.shape
    pp = np.asarray(pp, dtype=np.float64)
    rng = np.random.default_rng(seed)
    S = sigma * rng.standard_normal(n)
    # Bernoulli race: neuron j fires at sample t iff it is the first j with pp[j] > u[t, j].
    # Draw the race for every sample, then drop candidates that fall inside an active template
    # (utils.jl:63 only races while active_cell == 0).
    cand_t = []
    cand_j = []
    block = 1 << 20
    for t0 in range(0, n, block):
        m = min(block, n - t0)
        u = rng.random((m, ncells))
        hit = pp[None, :] > u
        anyhit = hit.any(axis=1)
        tt = np.nonzero(anyhit)[0]
        cand_t.append(tt + t0)
        cand_j.append(hit[tt].argmax(axis=1))
    cand_t = np.concatenate(cand_t)
    cand_j = np.concatenate(cand_j)
    onsets = []
    busy_until = -1  # last sample index occupied by the running template
    for t, j in zip(cand_t.tolist(), cand_j.tolist()):
        if t <= busy_until:
            continue
        m = min(nstates, n - t)
        S[t:t + m] += templates[:m, j]
        busy_until = t + nstates - 1
        onsets.append((t, j))
    if return_states:
        return S, onsets
    return S
