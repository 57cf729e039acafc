"""BASELINE configs 4 and 5 at their workload size on one MI355X, through size-independent properties:

* config 4 (N=8 templates x K=128 states, 64 channels x 10 M samples on 8 GPUs): one GPU's share =
  8 channels x 10 M samples, decode + E-step per channel; the first channel is the input on which the
  round-1 lane-per-chain engine missed its own boundary certificate (bench.py --neurons 8 --states 128,
  seed 1234: every second sample lies inside a spike);
* config 5 (N=16 x K=256 -> 4081 states, and the "4097-state" reading K=257; 100 M samples per channel).

Checked: every boundary certificate (Viterbi diag[0], forward diag[3], backward diag[5]) is zero; every
near-tie decision the sweep flagged on the decoded path was re-decided with the reference's serial
arithmetic (wave_ties.hip; diag[7] = decisions left unresolved = 0, the counts are printed); the decoded
path equals the op-for-op strict engine's -- over the WHOLE 10 M samples for config 4's first channel,
on sampled windows otherwise; the posterior mass sums to T (every column of gamma is a distribution); the
path is a valid path of the model; and, on a short signal of the same model shape, the E-step equals the
CPU oracle's (baumwelch.jl:205-309)."""
import numpy as np
import pytest

from conftest import to_oracle_sm

pytestmark = pytest.mark.gpu


def bench_model(H, N, K):
    """the model/signal family of bench.py (SURVEY 8d synthetic inputs scaled to N templates, K states)"""
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    pp = [[0.003, 0.001, 0.002, 0.0015][i % 4] * (60.0 / K) for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    return temps, pp, sm


def check_valid_path(x, K):
    L = K - 1
    d = np.diff(x.astype(np.int64))
    ring = x[:-1] > 1
    last = ring & (((x[:-1] - 2) % L) == L - 1)
    assert np.all(d[ring & ~last] == 1)
    nxt = x[1:][last]
    assert np.all((nxt == 1) | (((nxt - 2) % L) == 0))


def strict_windows(H, y, x, sm, temps, sigma, starts, length, margin):
    """decode windows with the strict engine (bit-exact by construction) and compare their interiors:
    a window starts and ends where the decoded path is silent, so its own first/last columns do not matter"""
    H.set_option("engine", H.ENGINE_STRICT)
    try:
        for s in starts:
            a, b = s, s + length
            while x[a] != 1:
                a -= 1
            while x[b] != 1:
                b += 1
            xs, _ = H.viterbi(y[a:b + 1], sm, temps, sigma)
            lo, hi = margin, (b + 1 - a) - margin
            assert np.array_equal(xs[lo:hi], x[a + lo:a + hi]), "window at %d differs from the strict engine" % s
    finally:
        H.set_option("engine", H.ENGINE_AUTO)


def decode_estep(H, plan, y, N, K):
    import torch
    T = len(y)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    dx = torch.zeros(T, dtype=torch.int16, device="cuda")
    dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    plan.decode_estep(dy, dx, dll, stats, st)
    diag = plan.diagnostics(st)
    ties = plan.tie_stats(st)
    x = dx.cpu().numpy()
    s = stats.cpu().numpy()
    del dy, dx
    return x, float(dll.cpu()[0]), s, diag, ties


def test_config4_one_gpu_share_8_channels_x_10M(H):
    N, K, T = 8, 128, 10_000_000
    temps, pp, sm = bench_model(H, N, K)
    NL = N * (K - 1)
    H.set_option("engine", H.ENGINE_AUTO)
    plan = H.Plan(T, sm, temps, 0.3)
    assert plan.info()["engine"] == H.ENGINE_WAVE
    near_ties = []
    keep = None
    try:
        for ch in range(8):
            y = H.create_signal(T, 0.3, pp, temps, seed=1234 + ch)
            x, ll, s, diag, ties = decode_estep(H, plan, y, N, K)
            assert diag[0] == 0 and diag[3] == 0 and diag[5] == 0, (ch, diag)
            # flagged decisions on the path (margin inside the reference's own rounding at |T1| ~ 2e6): all settled
            assert diag[7] == 0 and ties["unresolved"] == 0, (ch, diag, ties)
            near_ties.append((ties["flagged"], ties["decided"], ties["flips"], ties["longest_walk"]))
            assert max(diag[4], diag[6]) < 1e-9
            assert abs(s[:NL].sum() + s[3 * NL + N] - T) < 1e-7 * T          # posterior mass = T
            check_valid_path(x, K)
            assert np.mean(x > 1) > 0.4                                       # the busy regime: half of all samples inside spikes
            if ch == 0:
                keep = (y, x)
            elif ch == 1:
                strict_windows(H, y, x, sm, temps, 0.3, [1_000_000, 8_765_432], 20_000, 2_000)
    finally:
        plan.close()
    print("config 4 (flagged, re-decided, flips, longest walk) per channel:", near_ties)
    # the first channel over its whole length against the op-for-op sweep (1017 x 1e7 Int16 back-pointers = 20 GB)
    y, x = keep
    H.set_option("engine", H.ENGINE_STRICT)
    try:
        xs, lls = H.viterbi(y, sm, temps, 0.3)
    finally:
        H.set_option("engine", H.ENGINE_AUTO)
        H.shutdown()
    nbad = int(np.count_nonzero(xs != x))
    assert nbad == 0, "config 4 channel 0 differs from the strict engine at %d samples, first at %d" % (
        nbad, int(np.argmax(xs != x)))


@pytest.mark.parametrize("K,T", [(256, 100_000_000), (257, 100_000_000)])
def test_config5_long_channel(H, K, T):
    N = 16
    temps, pp, sm = bench_model(H, N, K)
    assert sm.nstates == 1 + N * (K - 1)           # 4081 / 4097
    NL = N * (K - 1)
    y = H.create_signal(T, 0.3, pp, temps, seed=4321)
    H.set_option("engine", H.ENGINE_AUTO)
    plan = H.Plan(T, sm, temps, 0.3)
    assert plan.info()["engine"] == H.ENGINE_WAVE
    try:
        x, ll, s, diag, ties = decode_estep(H, plan, y, N, K)
    finally:
        plan.close()
    assert diag[0] == 0 and diag[3] == 0 and diag[5] == 0, diag
    # near-ties: at t ~ 1e8 the reference's trellis values are ~2e7 (ulp 4e-9), and a 100 M-sample channel
    # with ~7e5 spikes holds decisions whose margin is inside the worst case of that noise:
    # each is re-decided with the reference's serial arithmetic, none may stay open
    print("config 5 (K=%d, T=%d) near-tie decisions on the path: %s" % (K, T, ties))
    assert diag[7] == 0 and ties["unresolved"] == 0, (diag, ties)
    assert max(diag[4], diag[6]) < 1e-9
    assert abs(s[:NL].sum() + s[3 * NL + N] - T) < 1e-7 * T
    check_valid_path(x, K)
    strict_windows(H, y, x, sm, temps, 0.3, [3_000_000, T - 2_000_000], 12_000, 1_500)


@pytest.mark.parametrize("N,K", [(16, 256), (8, 128)])
def test_config_shape_estep_matches_oracle(O, H, N, K):
    # the E-step of the config 4 / 5 model shapes against the oracle's update() on a short signal
    T = 9_000
    temps, pp, sm = bench_model(H, N, K)
    y = H.create_signal(T, 0.3, np.array(pp) * 4, temps, seed=77)
    rng = np.random.default_rng(1)
    mu = np.asfortranarray(temps * rng.uniform(0.8, 1.1, N)[None, :])
    mu[0, :] = 0
    H.set_option("engine", H.ENGINE_WAVE)
    try:
        sm_n, mu_n, sig_n = H.train_step(y, sm, mu.copy(order="F"), 0.4)
    finally:
        H.set_option("engine", H.ENGINE_AUTO)
    osm_n, omu, osig, olp, opp = O.train_step(y, to_oracle_sm(O, sm), mu.copy(order="F"), 0.4)
    assert np.allclose(mu_n, omu, rtol=1e-8, atol=1e-11), np.abs(mu_n - omu).max()
    assert abs(sig_n - osig) <= 1e-8 * osig
    assert np.allclose(sm_n.transitions["lp"], osm_n.val, rtol=1e-8, atol=1e-12)
