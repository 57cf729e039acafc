"""Host-side mirror of the reference's API for the hot path.

Julia is not available in the build image, so the host side above the C ABI is written in Python
with the reference's names, argument meaning and error behaviour; julia/HMMSpikeSorterHIP.jl holds
the equivalent `ccall` overrides for a Julia host.  Everything numeric happens behind
libhmmsort_hip.so on the GPU; this module only marshals arrays (column-major, 1-based ids, as
Julia lays them out).

Reference methods mirrored (file:line relative to the reference root):
  StateMatrix(N,K,lp,allow_overlaps) / StateMatrix(states,pp,K,lp)   src/types.jl:135-151
  forward / backward / update / train_model                           src/baumwelch.jl:25,73,205,311-370
  viterbi                                                             src/viterbi.jl:44
  reconstruct_signal                                                  src/reconstruction.jl:1
  unroll_mlseq                                                        src/extraction.jl:4
  fit(HMMSpikingModel, templates, X, chunksize)                       src/fit.jl:11-42
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import TRANS_DTYPE, HmmsortError, check, lib, ptr
from .synth import create_spike_template


@dataclass
class StateMatrix:
    """src/types.jl:1-9.  `transitions` is a structured array of (src, dst, lp) 24-byte records,
    binary-identical to Julia's Vector{Tuple{Int64,Int64,Float64}}."""
    states: np.ndarray        # N x S int16 (Fortran order), entry = 1-based row of mu
    transitions: np.ndarray   # R records, reference order
    pi: np.ndarray            # S float64, never read by the hot path
    K: int
    N: int
    nstates: int
    resolve_overlaps: bool = True

    def isempty(self):        # types.jl:13
        return self.states.size == 0

    @staticmethod
    def null():
        """StateMatrix()  types.jl:12: the null model, a single noise state"""
        tr = np.zeros(1, dtype=TRANS_DTYPE)
        tr[0] = (1, 1, 0.0)
        return StateMatrix(np.ones((1, 1), dtype=np.int16, order="F"), tr, np.array([1.0]), 0, 0, 1, False)

    @staticmethod
    def create(N, K, lp, allow_overlaps=True, pp=None):
        """StateMatrix(N, K, lp[, pp], allow_overlaps=true)  types.jl:135-146."""
        L = lib()
        S = L.hmmsort_generate_states(int(N), int(K), int(bool(allow_overlaps)), None)
        if S < 0:
            raise HmmsortError(int(S), _lib.last_error())
        states = np.zeros((N, S), dtype=np.int16, order="F")
        L.hmmsort_generate_states(int(N), int(K), int(bool(allow_overlaps)), ptr(states))
        if pp is None:
            pp = np.log(np.ones(S) / S)
        return StateMatrix.from_states(states, pp, K, lp, allow_overlaps)

    @staticmethod
    def from_states(states1, pp, K, lp, allow_overlaps=True):
        """StateMatrix(states, pp, K, lp; allow_overlaps)  types.jl:148-151, with the transition
        list produced in closed form (hmmsort_build_transitions) instead of the all-pairs scan.
        `states1` is the 1-based table (the reference passes `lA.states .- 1`, baumwelch.jl:265)."""
        L = lib()
        states1 = np.asfortranarray(states1, dtype=np.int16)
        N, S = states1.shape
        lp = np.ascontiguousarray(lp, dtype=np.float64)
        R = L.hmmsort_build_transitions(N, int(K), ptr(lp), len(lp), int(bool(allow_overlaps)),
                                        None, 0)
        if R < 0:
            raise HmmsortError(int(R), _lib.last_error())
        tr = np.zeros(R, dtype=TRANS_DTYPE)
        L.hmmsort_build_transitions(N, int(K), ptr(lp), len(lp), int(bool(allow_overlaps)),
                                    ptr(tr), R)
        return StateMatrix(states1, tr, np.array(pp, dtype=np.float64), int(K), int(N), int(S),
                           bool(allow_overlaps))


@dataclass
class HMMSpikeTemplateModel:   # types.jl:15-19
    state_matrix: StateMatrix
    mu: np.ndarray
    sigma: float


@dataclass
class HMMSpikingModel:         # types.jl:21-26
    template_model: HMMSpikeTemplateModel
    ml_seq: np.ndarray
    ll: float
    y: np.ndarray


def _model_args(lA, mu, sigma):
    mu = np.asfortranarray(mu, dtype=np.float64)
    if mu.ndim != 2 or mu.shape != (lA.K, lA.N):
        raise ValueError("mu must be K x N = %d x %d, got %s" % (lA.K, lA.N, mu.shape))
    tr = np.ascontiguousarray(lA.transitions, dtype=TRANS_DTYPE)
    st = np.asfortranarray(lA.states, dtype=np.int16)
    keep = (st, tr, mu)
    return keep, (ptr(st), lA.N, lA.K, lA.nstates, ptr(tr), len(tr), ptr(mu), float(sigma))


def _signal(y):
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("signal must be one-dimensional")
    return y


def viterbi(y, lA, mu, sigma):
    """viterbi(y, lA::StateMatrix, mu, sigma) -> (x::Vector{Int16}, ll)   viterbi.jl:44-98."""
    raw = isinstance(y, np.ndarray) and y.dtype == np.int16 and y.ndim == 1
    # int16 acquisition samples cross PCIe as they are and are widened in HBM (hmmsort.jl:84-88)
    y = np.ascontiguousarray(y) if raw else _signal(y)
    keep, margs = _model_args(lA, mu, sigma)
    x = np.zeros(len(y), dtype=np.int16)
    ll = C.c_double(0.0)
    entry = lib().hmmsort_viterbi_i16 if raw else lib().hmmsort_viterbi
    check(entry(ptr(y), len(y), *margs, ptr(x), C.cast(C.byref(ll), C.c_void_p)))
    if _lib.get_option("last_escalations") < 0:
        import warnings
        warnings.warn(_lib.last_error())      # near-ties the strict sweep could not re-decide (table too large)
    return x, ll.value


def forward(V, lA, mu, sigma):
    """forward(V, lA, mu, sigma) -> alpha (S x T)   baumwelch.jl:25-51."""
    V = _signal(V)
    keep, margs = _model_args(lA, mu, sigma)
    a = np.empty((lA.nstates, len(V)), dtype=np.float64, order="F")
    check(lib().hmmsort_forward(ptr(V), len(V), *margs, ptr(a)))
    return a


def backward(V, lA, mu, sigma):
    """backward(V, lA, mu, sigma) -> beta (S x T)   baumwelch.jl:73-98."""
    V = _signal(V)
    keep, margs = _model_args(lA, mu, sigma)
    b = np.empty((lA.nstates, len(V)), dtype=np.float64, order="F")
    check(lib().hmmsort_backward(ptr(V), len(V), *margs, ptr(b)))
    return b


def _finish_step(lA, mu, sig, lp, nlp, pp):
    lp_new = lp[: nlp.value].copy()
    # baumwelch.jl:265: StateMatrix(lA.states .- 1, pp, K, xb[2:end]; allow_overlaps)
    lA_new = StateMatrix.from_states(lA.states, pp, lA.K, lp_new, lA.resolve_overlaps)
    return lA_new, mu, sig.value


def update(alpha, beta, lA, mu, sigma, x):
    """update(alpha, beta, lA, mu, sigma, x) -> (StateMatrix, mu, sigma)   baumwelch.jl:205-309.
    `mu` is overwritten in place (as the reference's fill!(mu, 0.0) does) when it is a
    Fortran-ordered float64 array; it is also returned."""
    x = _signal(x)
    alpha = np.asfortranarray(alpha, dtype=np.float64)
    beta = np.asfortranarray(beta, dtype=np.float64)
    if alpha.shape != (lA.nstates, len(x)) or beta.shape != alpha.shape:
        raise ValueError("alpha/beta must be S x T")
    mu_in = mu
    keep, margs = _model_args(lA, mu, sigma)
    st, tr, mu_f = keep
    mu_f = mu_f if mu_f is not mu_in else mu_in  # in place when the layout allows
    sig = C.c_double(0.0)
    nlp = C.c_int64(0)
    lp = np.zeros(len(tr), dtype=np.float64)
    pp = np.zeros(lA.nstates, dtype=np.float64)
    check(lib().hmmsort_update(ptr(alpha), ptr(beta), ptr(x), len(x), ptr(st), lA.N, lA.K,
                               lA.nstates, ptr(tr), len(tr), ptr(mu_f), float(sigma),
                               C.cast(C.byref(sig), C.c_void_p), ptr(lp), len(lp), C.byref(nlp),
                               ptr(pp)))
    if mu_f is not mu_in and isinstance(mu_in, np.ndarray) and mu_in.shape == mu_f.shape:
        mu_in[...] = mu_f
    return _finish_step(lA, mu_f, sig, lp, nlp, pp)


def train_step(X, state_matrix, mu0, sigma0, verbose=0):
    """train_model(X, state_matrix, mu0, sigma0) = forward -> backward -> update, one EM step
    baumwelch.jl:362-370.  One C-ABI call; alpha/beta never leave the GPU."""
    X = _signal(X)
    keep, margs = _model_args(state_matrix, mu0, sigma0)
    st, tr, mu_f = keep
    mu_f = np.array(mu_f, dtype=np.float64, order="F", copy=True) if mu_f is mu0 else mu_f
    sig = C.c_double(0.0)
    nlp = C.c_int64(0)
    lp = np.zeros(len(tr), dtype=np.float64)
    pp = np.zeros(state_matrix.nstates, dtype=np.float64)
    check(lib().hmmsort_em_step(ptr(X), len(X), ptr(st), state_matrix.N, state_matrix.K,
                                state_matrix.nstates, ptr(tr), len(tr), ptr(mu_f), float(sigma0),
                                C.cast(C.byref(sig), C.c_void_p), ptr(lp), len(lp), C.byref(nlp),
                                ptr(pp)))
    if isinstance(mu0, np.ndarray) and mu0.shape == mu_f.shape and mu0.dtype == np.float64:
        mu0[...] = mu_f  # the reference updates the caller's mu in place (baumwelch.jl:268)
    return _finish_step(state_matrix, mu_f, sig, lp, nlp, pp)


class _EMSession:
    """EM steps on one signal that stays in HBM: the signal is uploaded once and one plan is
    re-armed with hmmsort_plan_set_model between steps (ring-engine models; anything else, a
    changed model shape or a failed warm-up certificate goes through train_step, which escalates)."""

    def __init__(self, X):
        self.X = X
        self.dX = None
        self.plan = None
        self.key = None

    def close(self):
        if self.plan is not None:
            self.plan.close()
            self.plan = None

    def step(self, lA, mu, sigma, verbose=0):
        import torch
        from . import _lib
        from .device import Plan
        key = (lA.N, lA.K, lA.nstates, len(lA.transitions), lA.resolve_overlaps)
        try:
            if self.key != key:                 # first step, or the model shape changed
                self.close()
                self.key = key
                self.plan = Plan(len(self.X), lA, mu, sigma)
                if self.plan.info()["engine"] not in (_lib.ENGINE_RING, _lib.ENGINE_WAVE):
                    self.close()                # remembered through self.key: not retried
            elif self.plan is not None:
                self.plan.set_model(lA, mu, sigma)
        except _lib.HmmsortError:
            self.close()
        if self.plan is None:
            return train_step(self.X, lA, mu, sigma, verbose=verbose)
        if self.dX is None:
            self.dX = torch.from_numpy(self.X).cuda()
            self.stats = torch.zeros(self.plan.stats_len(), dtype=torch.float64, device="cuda")
        out = torch.zeros(self.plan.mstep_len(), dtype=torch.float64, device="cuda")
        self.plan.estep(self.dX, self.stats)
        self.plan.mstep(self.stats, out)
        d = self.plan.diagnostics()
        if d[3] != 0 or d[5] != 0:
            return train_step(self.X, lA, mu, sigma, verbose=verbose)
        o = out.cpu().numpy()
        K, N = lA.K, lA.N
        mu_n = np.asfortranarray(o[:K * N].reshape((K, N), order="F"))
        if isinstance(mu, np.ndarray) and mu.shape == mu_n.shape and mu.dtype == np.float64:
            mu[...] = mu_n  # the reference updates the caller's mu in place (baumwelch.jl:268)
        lA_n = StateMatrix.from_states(lA.states, o[K * N + 1 + N:], K, o[K * N + 1:K * N + 1 + N],
                                       lA.resolve_overlaps)
        return lA_n, mu_n, float(o[K * N])


def train_model(X, *args, callback=None, verbose=0, p0=None, rng=None, postprocess="reference"):
    """The three `train_model` methods of baumwelch.jl:

      train_model(X, state_matrix, mu0, sigma0)                  one EM step         :362-370
      train_model(X, state_matrix, mu, sigma, nsteps[, callback]) EM loop             :324-354
      train_model(X, N=3, K=60, resolve_overlaps=False, nsteps=8[, callback])         :311-322

    The loop stays on the host exactly as in the reference (callback(mu) before every step, stop
    when the state matrix becomes empty); each step is one GPU call.  Between the two rounds of
    steps the reference merges and prunes templates (condense_templates, remove_sparse, remove_small,
    baumwelch.jl:340-349): `postprocess="reference"` (default) runs the restatement in postprocess.py,
    `None` skips the stage, a callable `postprocess(state_matrix, mu, sigma) -> (state_matrix, mu)`
    replaces it.
    """
    X = _signal(X)
    if len(args) >= 1 and isinstance(args[0], StateMatrix):
        state_matrix, mu, sigma = args[0], args[1], float(args[2])
        if len(args) == 3:
            return train_step(X, state_matrix, mu, sigma, verbose=verbose)
        nsteps = int(args[3])
        cb = args[4] if len(args) > 4 else callback
        mu = np.array(mu, dtype=np.float64, order="F", copy=True)
        em = _EMSession(X)
        try:
            for _ in range(nsteps):
                if cb is not None:
                    cb(mu)
                state_matrix, mu, sigma = em.step(state_matrix, mu, sigma, verbose=verbose)
                if state_matrix.isempty():
                    break
            if postprocess == "reference":
                from .postprocess import reference_postprocess
                state_matrix, mu = reference_postprocess(state_matrix, mu, sigma, verbose=verbose)
            elif postprocess is not None:
                state_matrix, mu = postprocess(state_matrix, mu, sigma)
            if state_matrix.N == 0:
                return state_matrix, mu, sigma      # every template was pruned: the null model
            for _ in range(nsteps // 2):
                state_matrix, mu, sigma = em.step(state_matrix, mu, sigma, verbose=verbose)
        finally:
            em.close()
        return state_matrix, mu, sigma
    # random initialisation, baumwelch.jl:311-322
    N = int(args[0]) if len(args) > 0 else 3
    K = int(args[1]) if len(args) > 1 else 60
    resolve_overlaps = bool(args[2]) if len(args) > 2 else False
    nsteps = int(args[3]) if len(args) > 3 else 8
    cb = args[4] if len(args) > 4 else callback
    if p0 is None:
        p0 = 2.0 ** (-3 * K / 2)
    if rng is None:
        rng = np.random.default_rng()
    lp = np.log(np.full(N, p0))
    state_matrix = StateMatrix.create(N, K, lp, resolve_overlaps)
    sigma = float(np.std(X, ddof=1))          # Julia's std is the corrected estimator
    mu = np.ones((K, N), order="F")
    for i in range(N):
        mu[:, i] = create_spike_template(K, 3 * sigma * rng.random(), 0.5 + 0.1 * rng.standard_normal(),
                                         1.5 * rng.random())
    mu[0, :] = 0.0
    return train_model(X, state_matrix, mu, sigma, nsteps, cb, verbose=verbose,
                       postprocess=postprocess)


def reconstruct_signal(x, lA, mu, sigma=None):
    """reconstruct_signal(x, lA, mu, sigma) -> Y2 (sigma unused)   reconstruction.jl:1-10."""
    x = np.ascontiguousarray(x, dtype=np.int16)
    mu = np.asfortranarray(mu, dtype=np.float64)
    st = np.asfortranarray(lA.states, dtype=np.int16)
    out = np.zeros(len(x), dtype=np.float64)
    check(lib().hmmsort_reconstruct(ptr(x), len(x), ptr(st), lA.N, lA.nstates, ptr(mu),
                                    mu.shape[0], ptr(out)))
    return out


def unroll_mlseq(mlseq, state_matrix):
    """unroll_mlseq(mlseq, state_matrix) -> N x T Int16   extraction.jl:4-13."""
    mlseq = np.ascontiguousarray(mlseq, dtype=np.int16)
    st = np.asfortranarray(state_matrix.states, dtype=np.int16)
    out = np.zeros((state_matrix.N, len(mlseq)), dtype=np.int16, order="F")
    check(lib().hmmsort_unroll_mlseq(ptr(mlseq), len(mlseq), ptr(st), state_matrix.N,
                                     state_matrix.nstates, ptr(out)))
    return out


def fit(templates, X, chunksize=None):
    """fit(HMMSpikingModel, templates, X[, chunksize])   fit.jl:6-9 and :11-42.

    With `chunksize` the signal is decoded in sequentially dependent chunks with the reference's
    stitch rule (leading non-silent samples of a chunk are skipped, trailing ones are handed to
    the next chunk, which restarts at the last silent sample).  The reference's call to the
    removed `gc()` (fit.jl:19) is not reproduced."""
    raw = isinstance(X, np.ndarray) and X.dtype == np.int16 and X.ndim == 1
    X = np.ascontiguousarray(X) if raw else _signal(X)
    lA, mu, sigma = templates.state_matrix, templates.mu, templates.sigma
    if chunksize is None:
        x, ll = viterbi(X, lA, mu, sigma)
        return HMMSpikingModel(templates, x, ll, X)
    n = len(X)
    i = j = 1
    ml_seq = np.ones(n, dtype=np.int16)
    ll = 0.0
    # The chunks depend on each other (a chunk restarts where the previous decode was last silent),
    # so they run one after the other -- but on the device: the signal is uploaded once and one
    # plan per chunk length is reused (the host-buffer entry point would rebuild its plan and copy
    # the chunk for every call).
    import torch
    from .device import Plan
    # duplicate templates: which twin the reference decodes hangs on the last bit of its own sums
    # (DESIGN 3.2); the host-buffer entry point knows (strict engine), the plan API does not
    muf = np.asarray(mu, dtype=np.float64)
    twins = any(np.array_equal(muf[:, a], muf[:, b]) for a in range(lA.N) for b in range(a + 1, lA.N))
    dX = torch.from_numpy(X).cuda()
    if raw:
        # the acquisition's int16 samples: 2 bytes per sample over PCIe, widened in HBM (hmmsort.jl:84-88)
        draw, dX = dX, torch.empty(n, dtype=torch.float64, device="cuda")
        check(lib().hmmsort_samples_to_f64(draw.data_ptr(), 0, n, 1, dX.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        del draw
    dx = torch.zeros(min(chunksize, n), dtype=torch.int16, device="cuda")
    dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    plans = {}
    try:
        while j < n:
            j = min(i + chunksize - 1, n)
            k = j - i + 1
            l = 1
            if k not in plans:
                plans[k] = Plan(k, lA, mu, sigma)
            plan = plans[k]
            if twins and plan.info()["engine"] == _lib.ENGINE_RING:
                # the lane-per-chain ring engine has no run-time near-tie detector (the wave and
                # blocked engines count near-ties in diag[7])
                x, _ll = viterbi(X[i - 1:j], lA, mu, sigma)
                plan = None
            else:
                plan.viterbi(dX.data_ptr() + (i - 1) * 8, dx, dll)
            dg = plan.diagnostics() if plan is not None else None
            if plan is None:
                pass
            elif dg[0] != 0 or (plan.info()["engine"] in (_lib.ENGINE_BLOCKED, _lib.ENGINE_WAVE) and dg[7] != 0):  # failed check / near-ties:
                x, _ll = viterbi(X[i - 1:j], lA, mu, sigma)   # the escalating entry point
            else:
                x, _ll = dx[:k].cpu().numpy(), float(dll.cpu()[0])
            if i > 1:
                while x[l - 1] > 1:
                    l += 1
            if j < n:
                while x[k - 1] > 1:
                    j -= 1
                    k -= 1
            ml_seq[i + l - 2:j] = x[l - 1:k]
            ll += _ll
            if j <= i:
                raise RuntimeError("chunk without a silent sample: the reference loops forever here")
            i = j
    finally:
        for pl in plans.values():
            pl.close()
    return HMMSpikingModel(templates, ml_seq, ll, X)


def predict(model):
    """StatsBase.predict(model)   fit.jl:54-56."""
    tm = model.template_model
    return reconstruct_signal(model.ml_seq, tm.state_matrix, tm.mu, tm.sigma)


def extract_spiketimes(model):
    """extract_spiketimes(model::HMMSpikingModel) -> one array of spike sample indices per neuron
    (1-based like the reference's findin result)   extraction.jl:15-24."""
    tm = model.template_model
    lA = tm.state_matrix
    x = np.ascontiguousarray(model.ml_seq, dtype=np.int16)
    mu = np.asfortranarray(tm.mu, dtype=np.float64)
    st = np.asfortranarray(lA.states, dtype=np.int16)
    cap = max(1, len(x) // 8)
    while True:
        times = np.zeros((lA.N, cap), dtype=np.int64)
        counts = np.zeros(lA.N, dtype=np.int64)
        check(lib().hmmsort_extract_spiketimes(ptr(x), len(x), ptr(st), lA.N, lA.nstates, ptr(mu),
                                               mu.shape[0], ptr(times), cap, ptr(counts)))
        if counts.max(initial=0) <= cap:
            return [times[i, :counts[i]].copy() for i in range(lA.N)]
        cap = int(counts.max())
