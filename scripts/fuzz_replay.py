"""Replay of ONE case of scripts/fuzz_gpu.py (same RNG draws, no GPU work for the cases before it), then its chunked fit
chunk by chunk through the plan API, structured and generic sweeps, against the oracle.  This is how the stale-operand bug
of the multi sweep's junction was located (fuzz_gpu.py 500 31, case 403; DESIGN 5, lesson vii).
usage: python scripts/fuzz_replay.py <case> <seed>"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm
O.build()
ONLY, SEED = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(SEED)
for case in range(ONLY + 1):
    ov = bool(rng.integers(0, 2))
    N = int(rng.integers(1, 5 if ov else 7))
    K = int(rng.integers(2, 34 if ov else 70))
    smax = 6000
    if ov and 1 + N * (K - 1) + N * (N - 1) // 2 * (K - 1) ** 2 > smax:
        K = max(2, int(np.sqrt(smax / max(1, N * (N - 1) // 2))))
    T = int(rng.integers(300, 40000))
    sigma = float(rng.uniform(0.15, 0.6))
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, rng.uniform(1.5, 5), rng.uniform(0.2, 1.0),
                                                                rng.uniform(0.1, 0.4)) for _ in range(N)], 1))
    pp = rng.uniform(5e-4, 8e-3, N) * min(1.0, 30.0 / K)
    yseed = int(rng.integers(1, 1 << 30))
    S = 1 + N * (K - 1) + (N * (N - 1) // 2 * (K - 1) ** 2 if ov else 0)
    blk = int(rng.choice([0, 0, 128, 192, 256, 512, 1024]))
    hal = int(rng.choice([0, 0, 64, 128, 256, 512]))
    cs = None
    if T >= 3000 and rng.random() < 0.5:
        cs = int(rng.integers(1000, T // 2))
    if T * S <= 4_000_000 and T >= 2:
        rng.uniform(0.8, 1.2, N)
print("case", ONLY, "N", N, "K", K, "ov", ov, "S", S, "T", T, "sigma", sigma, "blk", blk, "hal", hal, "cs", cs)
if cs is None:
    print("this case has no chunked fit"); sys.exit(0)
y = H.create_signal(T, sigma, pp, temps, seed=yseed)
sm = H.StateMatrix.create(N, K, np.log(pp), ov)
osm = to_oracle_sm(O, sm)
H.set_option("block", blk); H.set_option("halo", hal)
rc, ml, llc = O.fit_chunked(y, osm, temps, sigma, cs)
mdl = H.fit(H.HMMSpikeTemplateModel(sm, temps, sigma), y, cs)
bad = np.nonzero(mdl.ml_seq != ml)[0]
print("fit mismatches", len(bad), bad[:10], "ll", mdl.ll, llc)
# chunk by chunk with the plan API
import torch
i, n = 1, T
st = torch.cuda.current_stream().cuda_stream
dX = torch.from_numpy(y).cuda()
while True:
    j = min(i + cs - 1, n); k = j - i + 1
    seg = y[i - 1:j]
    xo, llo = O.viterbi(seg, osm, temps, sigma)
    for mode in ("multi", "generic"):
        if mode == "generic": os.environ["HMMSORT_PAIR"] = "0"
        else: os.environ.pop("HMMSORT_PAIR", None)
        plan = H.Plan(k, sm, temps, sigma)
        dx = torch.zeros(k, dtype=torch.int16, device="cuda"); dll = torch.zeros(1, dtype=torch.float64, device="cuda")
        plan.viterbi(dX.data_ptr() + (i - 1) * 8, dx, dll, st); torch.cuda.synchronize()
        d = plan.diagnostics(st); info = plan.info(); plan.close()
        x = dx.cpu().numpy()
        nb = int((x != xo).sum())
        print("chunk [%d,%d) %s: mismatches %d first %s diag0 %d ties %d block %d halo %d" % (i - 1, j, mode, nb, np.nonzero(x != xo)[0][:5], d[0], d[7], info["block"], info["halo"]))
    os.environ.pop("HMMSORT_PAIR", None)
    x = xo
    if j >= n: break
    while x[k - 1] > 1:
        j -= 1; k -= 1
    if j <= i: break
    i = j
