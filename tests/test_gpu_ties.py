"""The wave engine's exact near-tie resolver (csrc/wave_ties.hip) against the CPU oracle.

The reference decides `T1[k,i-1] + lp > T1[j,i]` (viterbi.jl:74-84) on an unnormalised trellis, so a
decision between candidates closer than the rounding of an O(t) serial sum is settled by that rounding.
The time-parallel sweep flags such decisions and the resolver replays the reference's own arithmetic for
the flagged ones on the decoded path.  These tests force the mechanism to work hard:

* `tie_scale` multiplies the flag threshold, so that ordinary decisions are flagged by the thousand: every
  one must then be re-decided to the oracle's answer (the path stays the oracle's bit for bit);
* the exact prefix T1[x_u, u] the resolver builds from per-block increments must equal a plain serial fold
  in the reference's operation order, bit for bit, at every block start;
* duplicate templates: every spike is a tie between the twins up to the last bits of the reference's own
  sums -- the decoded path must still be the oracle's, at trellis magnitudes (|T1| ~ 5e5) where per-chain
  frames cannot reproduce those bits;
* a recording that ends inside a twin spike: the final arg-max (viterbi.jl:90) is a near-tie.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import four_templates, to_oracle_sm

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def wave_engine(H):
    H.set_option("engine", H.ENGINE_WAVE)
    yield
    H.set_option("engine", H.ENGINE_AUTO)
    H.set_option("tie_scale", 1)
    H.set_option("tie_debug", 0)


def _decode(H, y, sm, mu, sigma, want_prefix=False):
    import torch
    plan = H.Plan(len(y), sm, mu, sigma)
    try:
        st = torch.cuda.current_stream().cuda_stream
        dy = torch.from_numpy(np.ascontiguousarray(y)).cuda()
        dx = torch.zeros(len(y), dtype=torch.int16, device="cuda")
        dll = torch.zeros(1, dtype=torch.float64, device="cuda")
        plan.viterbi(dy, dx, dll, st)
        diag = plan.diagnostics(st)
        ties = plan.tie_stats(st)
        x, ll = dx.cpu().numpy(), float(dll.cpu()[0])
        tv = None
        if want_prefix:
            nblk = (len(y) + 511) // 512
            tv = np.zeros(nblk + 1)
            fn = H._lib.lib().hmmsort_plan_debug_array
            fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
            fn.restype = C.c_int
            H._lib.check(fn(plan._h, 7, tv.ctypes.data_as(C.c_void_p), len(tv)))
    finally:
        plan.close()
    return x, ll, diag, ties, tv


def _bench_family(H, N, K, rng):
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    pp = rng.uniform(5e-4, 3e-3, N) * min(1.0, 60.0 / K) * min(1.0, 4.0 / N)
    return temps, pp


@pytest.mark.parametrize("N,K,T,seed,scale", [
    (4, 60, 200_000, 1, 30_000_000),
    (4, 60, 200_000, 2, 300_000_000),       # threshold ~ 4: almost every decision near a spike is flagged
    (3, 60, 20_000, 3, 1_000_000_000),
    (8, 128, 120_000, 4, 30_000_000),
    (16, 33, 60_000, 5, 100_000_000),
    (16, 256, 40_000, 6, 30_000_000),
    (1, 40, 30_011, 7, 100_000_000),
])
def test_forced_flags_are_redecided_to_the_oracle(O, H, N, K, T, seed, scale):
    rng = np.random.default_rng(seed)
    temps, pp = _bench_family(H, N, K, rng)
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, 0.3)
    H.set_option("tie_scale", scale)
    x, ll, diag, ties, _ = _decode(H, y, sm, temps, 0.3)
    print("N=%d K=%d T=%d scale=%g: %s" % (N, K, T, scale, ties))
    assert ties["flagged"] > (20 if T > 50_000 else 0), ties    # the mechanism was exercised
    assert ties["decided"] >= min(ties["flagged"], 4096) * 0.5
    assert ties["unresolved"] == 0 and diag[7] == 0 and diag[0] == 0, (ties, diag)
    nbad = int(np.count_nonzero(x != xo))
    assert nbad == 0, "path differs at %d samples, first at %d (%s)" % (nbad, int(np.argmax(x != xo)), ties)
    assert abs(ll - llo) <= 1e-9 * abs(llo)


def _serial_fold(y, x, sm, mu, sigma):
    """T1[x_u, u] along a path with the reference's operations (viterbi.jl:55-63,79,86), in Python floats"""
    import math
    tr = sm.transitions
    lp = {(int(a), int(b)): float(c) for a, b, c in zip(tr["src"], tr["dst"], tr["lp"])}
    st = np.asarray(sm.states)
    N = st.shape[0]
    mean = np.zeros(st.shape[1])
    for j in range(st.shape[1]):
        m = 0.0
        for l in range(N):
            m += mu[st[l, j] - 1, l]
        mean[j] = m
    A = -0.9189385332046727 - math.log(sigma)
    den = 2 * (sigma * sigma)
    out = np.zeros(len(y))
    if x[0] != 1:
        dd = float(y[0]) - mean[x[0] - 1]
        out[0] = A - (dd * dd) / den
    v = float(out[0])
    for u in range(1, len(y)):
        dd = float(y[u]) - float(mean[x[u] - 1])
        v = (v + lp[(int(x[u - 1]), int(x[u]))]) + (A - (dd * dd) / den)
        out[u] = v
    return out


@pytest.mark.parametrize("N,K,T,seed", [(4, 60, 300_000, 11), (2, 25, 150_001, 12)])
def test_exact_prefix_equals_the_serial_fold(H, N, K, T, seed):
    rng = np.random.default_rng(seed)
    temps, pp = _bench_family(H, N, K, rng)
    y = H.create_signal(T, 0.3, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    H.set_option("tie_scale", 30_000_000)
    H.set_option("tie_debug", 1)
    x, ll, diag, ties, tv = _decode(H, y, sm, temps, 0.3, want_prefix=True)
    assert ties["flagged"] > 0 and ties["unresolved"] == 0
    ref = _serial_fold(y, x, sm, temps, 0.3)
    nblk = (T + 511) // 512
    idx = np.minimum(np.arange(nblk + 1) * 512, T - 1)
    got, want = tv[:nblk + 1], ref[idx]
    # the last entry sits at block start nblk * 512 >= T only when T is a multiple of 512: compare what exists
    upto = nblk + 1 if nblk * 512 <= T - 1 else nblk
    bad = np.nonzero(got[:upto] != want[:upto])[0]
    assert len(bad) == 0, "exact prefix differs at block %d: %r vs %r" % (bad[0], got[bad[0]], want[bad[0]])
    # most blocks took the increment path, a few (binade crossings) were folded serially
    print("blocks %d, folded serially %d" % (nblk, ties["serial_blocks"]))
    assert ties["serial_blocks"] < nblk // 3
    # and ll is the descending sum of those values (viterbi.jl:92-96) to 1e-12
    assert abs(ll - float(np.sum(ref[1:][::-1]))) <= 1e-11 * abs(ll)


def _twin_model(H, K, pp_twin, extra=True):
    t1 = H.create_spike_template(K, 3.0, 0.8, 0.2)
    t2 = H.create_spike_template(K, 4.0, 0.3, 0.2)
    cols = [t1, t1.copy(), t2] if extra else [t1, t1.copy()]
    temps = np.asfortranarray(np.stack(cols, 1))
    pp = np.array([pp_twin, pp_twin, 0.001] if extra else [pp_twin, pp_twin])
    return temps, pp


@pytest.mark.parametrize("T,seed", [(400_000, 21), (2_000_000, 22)])
def test_duplicate_templates_decode_like_the_reference(O, H, T, seed):
    K = 60
    temps, pp = _twin_model(H, K, 0.002)
    # spikes of the twin template only from ring 0 of the generator; the decoder cannot tell the twins apart
    y = H.create_signal(T, 0.3, [0.003, 0.0, 0.001], temps, seed=seed)
    sm = H.StateMatrix.create(3, K, np.log(pp), False)
    xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, 0.3)
    x, ll, diag, ties, _ = _decode(H, y, sm, temps, 0.3)
    print("twins T=%d: %s" % (T, ties))
    assert ties["flagged"] > 100 and ties["unresolved"] == 0 and diag[7] == 0, ties
    nbad = int(np.count_nonzero(x != xo))
    assert nbad == 0, "path differs at %d samples, first at %d (%s)" % (nbad, int(np.argmax(x != xo)), ties)
    assert abs(ll - llo) <= 1e-9 * abs(llo)
    # the host entry point needs no strict fallback any more
    H.set_option("engine", H.ENGINE_AUTO)
    x2, ll2 = H.viterbi(y, sm, temps, 0.3)
    assert np.array_equal(x2, xo) and H.get_option("last_escalations") == 0


def test_recording_that_ends_inside_a_twin_spike(O, H):
    K, T = 60, 50_000
    temps, pp = _twin_model(H, K, 0.002, extra=False)
    y = H.create_signal(T, 0.3, [0.003, 0.0], temps, seed=31)
    # one more spike whose template is cut off by the end of the recording
    cut = 23
    y[T - cut:] += temps[1:cut + 1, 0]
    sm = H.StateMatrix.create(2, K, np.log(pp), False)
    xo, llo = O.viterbi(y, to_oracle_sm(O, sm), temps, 0.3)
    assert xo[-1] > 1, "the oracle should end inside the spike"
    x, ll, diag, ties, _ = _decode(H, y, sm, temps, 0.3)
    print("cut spike:", ties)
    assert ties["tail"] == 1 and ties["unresolved"] == 0
    assert np.array_equal(x, xo), int(np.count_nonzero(x != xo))


def test_batched_plan_resolves_each_channel(O, H):
    import torch
    K, T, C_ = 60, 120_000, 3
    temps, pp = _twin_model(H, K, 0.002)
    sm = H.StateMatrix.create(3, K, np.log(pp), False)
    ys = [H.create_signal(T, 0.3, [0.003, 0.0, 0.001], temps, seed=40 + i) for i in range(C_)]
    plan = H.Plan.batched(T, [sm] * C_, [temps] * C_, [0.3] * C_)
    try:
        st = torch.cuda.current_stream().cuda_stream
        dy = torch.from_numpy(np.stack(ys)).cuda()
        dx = torch.zeros((C_, T), dtype=torch.int16, device="cuda")
        dll = torch.zeros(C_, dtype=torch.float64, device="cuda")
        plan.viterbi(dy, dx, dll, st)
        ties = plan.tie_stats(st)
        diag = plan.diagnostics(st)
        x = dx.cpu().numpy()
    finally:
        plan.close()
    assert ties["unresolved"] == 0 and diag[7] == 0 and ties["flagged"] > 100
    for i in range(C_):
        xo, _ = O.viterbi(ys[i], to_oracle_sm(O, sm), temps, 0.3)
        assert np.array_equal(x[i], xo), (i, int(np.count_nonzero(x[i] != xo)))
