"""ctypes front-end of the CPU oracle (oracle/hmm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  Array conventions follow the reference:
column-major (Fortran order) matrices, 1-based state ids.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64, _f64 = C.c_int64, C.c_double
_pi16 = C.POINTER(C.c_int16)
_pi64 = C.POINTER(C.c_int64)
_pf64 = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libhmm_oracle.so")
    src = os.path.join(_HERE, "hmm_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "libhmm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib(path=None):
    global _LIB
    if _LIB is None or path is not None:
        so = path or build()
        L = C.CDLL(so)
        L.hmm_oracle_funcl3.restype = _f64
        L.hmm_oracle_funcl3.argtypes = [_f64, _f64, _f64]
        L.hmm_oracle_funcl4.restype = _f64
        L.hmm_oracle_funcl4.argtypes = [_f64, _f64, _f64, _f64]
        L.hmm_oracle_logsumexpl.restype = _f64
        L.hmm_oracle_logsumexpl.argtypes = [_f64, _f64]
        L.hmm_oracle_generate_states.restype = _i64
        L.hmm_oracle_generate_states.argtypes = [_i64, _i64, C.c_int, _pi16]
        L.hmm_oracle_isvalid_transition.restype = _f64
        L.hmm_oracle_isvalid_transition.argtypes = [_pi16, _i64, _i64, _pf64, _i64, _i64, _i64]
        L.hmm_oracle_get_valid_transitions.restype = _i64
        L.hmm_oracle_get_valid_transitions.argtypes = [_pi16, _i64, _i64, _i64, _pf64, _i64,
                                                       _pi64, _pi64, _pf64, _i64]
        model = [_pi16, _i64, _i64, _i64, _pi64, _pi64, _pf64, _i64, _pf64, _f64]
        L.hmm_oracle_forward.restype = C.c_int
        L.hmm_oracle_forward.argtypes = [_pf64, _i64] + model + [_pf64]
        L.hmm_oracle_backward.restype = C.c_int
        L.hmm_oracle_backward.argtypes = [_pf64, _i64] + model + [_pf64]
        L.hmm_oracle_update.restype = C.c_int
        L.hmm_oracle_update.argtypes = [_pf64, _pf64, _i64, _pi16, _i64, _i64, _i64, _pi64, _pi64,
                                        _pf64, _i64, _pf64, _f64, _pf64, _pf64, _pf64, _i64, _pf64,
                                        _pi64]
        L.hmm_oracle_viterbi.restype = C.c_int
        L.hmm_oracle_viterbi.argtypes = [_pf64, _i64] + model + [_pi16, _pf64, C.c_int, _pf64]
        L.hmm_oracle_reconstruct.restype = None
        L.hmm_oracle_reconstruct.argtypes = [_pi16, _i64, _pi16, _i64, _i64, _pf64, _i64, _pf64]
        L.hmm_oracle_unroll_mlseq.restype = None
        L.hmm_oracle_unroll_mlseq.argtypes = [_pi16, _i64, _pi16, _i64, _pi16]
        L.hmm_oracle_extract_spiketimes.restype = None
        L.hmm_oracle_extract_spiketimes.argtypes = [_pi16, _i64, _pi16, _i64, _i64, _pf64, _i64,
                                                    _pi64, _i64, _pi64]
        L.hmm_oracle_fit_chunked.restype = C.c_int
        L.hmm_oracle_fit_chunked.argtypes = [_pf64, _i64, _i64] + model + [_pi16, _pf64]
        if path is not None:
            return L
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


@dataclass
class StateMatrix:
    """types.jl:1-9 (field order kept; transitions split into three parallel arrays)."""
    states: np.ndarray      # N x S int16, Fortran order, 1-based rows of mu
    src: np.ndarray         # R int64, 1-based
    dst: np.ndarray         # R int64, 1-based
    val: np.ndarray         # R float64
    pi: np.ndarray          # S float64 (never read by the hot path)
    K: int
    N: int
    nstates: int
    resolve_overlaps: bool

    def _model_args(self, mu, sigma):
        return (_p(self.states, _pi16), self.N, self.K, self.nstates, _p(self.src, _pi64),
                _p(self.dst, _pi64), _p(self.val, _pf64), len(self.src), _p(mu, _pf64),
                float(sigma))


def generate_states(N, K, allow_overlaps=True):
    """types.jl:65-92; returns the 0-based table (N x S, Fortran order)."""
    L = lib()
    S = L.hmm_oracle_generate_states(N, K, int(allow_overlaps), None)
    st = np.zeros((N, S), dtype=np.int16, order="F")
    L.hmm_oracle_generate_states(N, K, int(allow_overlaps), _p(st, _pi16))
    return st


def state_matrix_from_states(states0, pp, K, lp, allow_overlaps=True):
    """types.jl:148-151  StateMatrix(states::Array{Int16,2}, pp, K, lp; allow_overlaps)."""
    L = lib()
    states0 = np.asfortranarray(states0, dtype=np.int16)
    N, S = states0.shape
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    R = L.hmm_oracle_get_valid_transitions(_p(states0, _pi16), N, S, K, _p(lp, _pf64), len(lp),
                                           None, None, None, 0)
    src = np.zeros(R, np.int64)
    dst = np.zeros(R, np.int64)
    val = np.zeros(R, np.float64)
    L.hmm_oracle_get_valid_transitions(_p(states0, _pi16), N, S, K, _p(lp, _pf64), len(lp),
                                       _p(src, _pi64), _p(dst, _pi64), _p(val, _pf64), R)
    return StateMatrix(np.asfortranarray(states0 + 1, dtype=np.int16), src, dst, val,
                       np.array(pp, dtype=np.float64), int(K), int(N), int(S), bool(allow_overlaps))


def state_matrix(N, K, lp, allow_overlaps=True, pp=None):
    """types.jl:135-146  StateMatrix(N, K, lp[, pp], allow_overlaps=true)."""
    st = generate_states(N, K, allow_overlaps)
    S = st.shape[1]
    if pp is None:
        pp = np.log(np.ones(S) / S)
    return state_matrix_from_states(st, pp, K, lp, allow_overlaps)


def _mu(mu):
    return np.asfortranarray(mu, dtype=np.float64)


def forward(V, sm, mu, sigma):
    V = np.ascontiguousarray(V, np.float64)
    mu = _mu(mu)
    a = np.empty((sm.nstates, len(V)), dtype=np.float64, order="F")
    rc = lib().hmm_oracle_forward(_p(V, _pf64), len(V), *sm._model_args(mu, sigma), _p(a, _pf64))
    assert rc == 0
    return a


def backward(V, sm, mu, sigma):
    V = np.ascontiguousarray(V, np.float64)
    mu = _mu(mu)
    a = np.empty((sm.nstates, len(V)), dtype=np.float64, order="F")
    rc = lib().hmm_oracle_backward(_p(V, _pf64), len(V), *sm._model_args(mu, sigma), _p(a, _pf64))
    assert rc == 0
    return a


def update(alpha, beta, sm, mu, sigma, x):
    """baumwelch.jl:205-309.  Returns (new StateMatrix, mu (new array), sigma, lp_new, pp)."""
    x = np.ascontiguousarray(x, np.float64)
    mu = _mu(mu).copy(order="F")
    alpha = np.asfortranarray(alpha)
    beta = np.asfortranarray(beta)
    sig = C.c_double(0.0)
    nt = C.c_int64(0)
    R = len(sm.src)
    xb = np.zeros(R, np.float64)
    pp = np.zeros(sm.nstates, np.float64)
    rc = lib().hmm_oracle_update(_p(alpha, _pf64), _p(beta, _pf64), len(x), _p(sm.states, _pi16),
                                 sm.N, sm.K, sm.nstates, _p(sm.src, _pi64), _p(sm.dst, _pi64),
                                 _p(sm.val, _pf64), R, _p(mu, _pf64), float(sigma), _p(x, _pf64),
                                 C.byref(sig), _p(xb, _pf64), R, _p(pp, _pf64), C.byref(nt))
    assert rc == 0
    lp_new = xb[: nt.value - 1].copy()
    sm_new = state_matrix_from_states(sm.states - 1, pp, sm.K, lp_new, sm.resolve_overlaps)
    return sm_new, mu, sig.value, lp_new, pp


def train_step(X, sm, mu, sigma):
    """baumwelch.jl:362-370  one EM step = forward -> backward -> update."""
    a = forward(X, sm, mu, sigma)
    b = backward(X, sm, mu, sigma)
    return update(a, b, sm, mu, sigma, X)


def viterbi(y, sm, mu, sigma, lean=True, return_T1=False):
    y = np.ascontiguousarray(y, np.float64)
    mu = _mu(mu)
    x = np.zeros(len(y), np.int16)
    ll = C.c_double(0.0)
    T1 = None
    if return_T1:
        lean = False
        T1 = np.empty((sm.nstates, len(y)), dtype=np.float64, order="F")
    rc = lib().hmm_oracle_viterbi(_p(y, _pf64), len(y), *sm._model_args(mu, sigma), _p(x, _pi16),
                                  C.byref(ll), int(lean), _p(T1, _pf64) if return_T1 else None)
    assert rc == 0
    if return_T1:
        return x, ll.value, T1
    return x, ll.value


def reconstruct_signal(x, sm, mu, sigma=None):
    x = np.ascontiguousarray(x, np.int16)
    mu = _mu(mu)
    out = np.zeros(len(x), np.float64)
    lib().hmm_oracle_reconstruct(_p(x, _pi16), len(x), _p(sm.states, _pi16), sm.N, sm.nstates,
                                 _p(mu, _pf64), mu.shape[0], _p(out, _pf64))
    return out


def unroll_mlseq(mlseq, sm):
    mlseq = np.ascontiguousarray(mlseq, np.int16)
    out = np.zeros((sm.N, len(mlseq)), dtype=np.int16, order="F")
    lib().hmm_oracle_unroll_mlseq(_p(mlseq, _pi16), len(mlseq), _p(sm.states, _pi16), sm.N,
                                  _p(out, _pi16))
    return out


def fit_chunked(X, sm, mu, sigma, chunksize):
    X = np.ascontiguousarray(X, np.float64)
    mu = _mu(mu)
    ml = np.zeros(len(X), np.int16)
    ll = C.c_double(0.0)
    rc = lib().hmm_oracle_fit_chunked(_p(X, _pf64), len(X), int(chunksize),
                                      *sm._model_args(mu, sigma), _p(ml, _pi16), C.byref(ll))
    return rc, ml, ll.value


def extract_spiketimes(ml_seq, sm, mu):
    """extraction.jl:15-24 -> list of 1-based sample-index arrays, one per neuron."""
    ml_seq = np.ascontiguousarray(ml_seq, np.int16)
    mu = _mu(mu)
    cap = len(ml_seq)
    times = np.zeros((sm.N, cap), np.int64)
    counts = np.zeros(sm.N, np.int64)
    lib().hmm_oracle_extract_spiketimes(_p(ml_seq, _pi16), len(ml_seq), _p(sm.states, _pi16), sm.N,
                                        sm.nstates, _p(mu, _pf64), mu.shape[0], _p(times, _pi64),
                                        cap, _p(counts, _pi64))
    return [times[i, :counts[i]].copy() for i in range(sm.N)]
