"""Randomised parity sweep of the 3-5 template overlap sweep (csrc/multi_sweep.hip): random template counts, ring
lengths, shapes incl. near-duplicates, firing rates, noise levels, silent means, block / warm-up requests, inserted
overlaps; short signals decode through hmmsort_viterbi (with its fallbacks) against the CPU oracle, long ones through
the plan API against the generic blocked sweep (HMMSORT_PAIR=0).  python scripts/fuzz_multi.py [n] [seed] [long]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from conftest import to_oracle_sm  # noqa: E402

O.build()


def make(rng, N, K, T, kind, sigma):
    temps = np.stack([H.create_spike_template(K, rng.uniform(1.5, 5), rng.uniform(0.2, 1.0), rng.uniform(0.1, 0.4))
                      for _ in range(N)], 1)
    if kind == "near_twin":
        temps[:, 1] = temps[:, 0] * (1 + 1e-9)
    if kind == "twin":
        temps[:, N - 1] = temps[:, 0]
    temps = np.asfortranarray(temps)
    pp = rng.uniform(5e-4, 6e-3, N) * min(1.0, 30.0 / K)
    if kind == "twin":
        pp[N - 1] = pp[0]
    y = H.create_signal(T, sigma, pp, temps, seed=int(rng.integers(1, 1 << 30)))
    L = K - 1
    for _ in range(int(rng.integers(0, 12))):
        t0 = int(rng.integers(L, T - 4 * L))
        d = int(rng.integers(0, L))
        a, b = rng.choice(N, 2, replace=False)
        y[t0:t0 + L] += temps[1:, a]
        y[t0 + d:t0 + d + L] += temps[1:, b]
        if rng.random() < 0.4:          # a third spike right after the first ends: pair exit -> new pair entry
            c = int(rng.choice([q for q in range(N) if q != a]))
            y[t0 + L:t0 + 2 * L] += temps[1:, c]
    mu = temps.copy(order="F")
    if kind == "silent_mean":
        mu[0, :] = rng.uniform(-0.05, 0.05, N)
    sm = H.StateMatrix.create(N, K, np.log(pp), True)
    return y, sm, mu


def case(rng, i, long_run):
    N = int(rng.choice([3, 3, 4, 4, 5]))
    if long_run:
        K = int(rng.integers(8, 61)) if N < 5 else int(rng.integers(8, 50))
        T = int(rng.integers(300_000, 1_200_000)) if N * K > 150 else int(rng.integers(1_000_000, 3_000_000))
    else:
        K = int(rng.integers(3, 22 if N < 5 else 16))
        T = int(rng.integers(4100, 30000))
    sigma = float(rng.uniform(0.15, 0.6))
    kind = rng.choice(["plain", "plain", "plain", "near_twin", "twin", "silent_mean"])
    y, sm, mu = make(rng, N, K, T, kind, sigma)
    H.set_option("block", int(rng.choice([0, 0, 128, 256, 512, 1024])))
    H.set_option("halo", int(rng.choice([0, 0, 64, 128, 256, 512])))
    tag = "case %d: N=%d K=%d T=%d sigma=%.2f %s" % (i, N, K, T, sigma, kind)
    try:
        if not long_run:
            x, ll = H.viterbi(y, sm, mu, sigma)
            esc = H.get_option("last_escalations")
            xo, llo = O.viterbi(y, to_oracle_sm(O, sm), mu, sigma)
            ok = np.array_equal(x, xo) and abs(ll - llo) <= 1e-9 * abs(llo)
            return ok, tag + " -> %s (esc %d)" % ("ok" if ok else "MISMATCH %d" % int((x != xo).sum()), esc)
        st = torch.cuda.current_stream().cuda_stream
        dy = torch.from_numpy(y).cuda()
        out = {}
        for mode in ("multi", "generic"):
            if mode == "generic":
                os.environ["HMMSORT_PAIR"] = "0"
            else:
                os.environ.pop("HMMSORT_PAIR", None)
            plan = H.Plan(T, sm, mu, sigma)
            dx = torch.zeros(T, dtype=torch.int16, device="cuda")
            dll = torch.zeros(1, dtype=torch.float64, device="cuda")
            plan.viterbi(dy, dx, dll, st)
            torch.cuda.synchronize()
            t0 = time.time()
            plan.viterbi(dy, dx, dll, st)
            torch.cuda.synchronize()
            dt = time.time() - t0
            out[mode] = (dx.cpu().numpy(), float(dll.cpu()[0]), plan.diagnostics(st), dt)
            plan.close()
        os.environ.pop("HMMSORT_PAIR", None)
        dp, dg = out["multi"][2], out["generic"][2]
        same = np.array_equal(out["multi"][0], out["generic"][0])
        ok = same or dp[7] > 0 or dg[7] > 0 or dp[0] > 0 or dg[0] > 0
        return ok, tag + " -> %s (multi diag0 %d ties %d %.1f Ms/s; generic diag0 %d ties %d %.1f Ms/s)" % (
            "same path" if same else ("differs, flagged" if ok else "MISMATCH unflagged %d" % int((out["multi"][0] != out["generic"][0]).sum())),
            dp[0], dp[7], T / out["multi"][3] / 1e6, dg[0], dg[7], T / out["generic"][3] / 1e6)
    except Exception as exc:  # noqa: BLE001
        return False, tag + " -> EXCEPTION %r" % (exc,)
    finally:
        H.set_option("block", 0)
        H.set_option("halo", 0)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    long_run = len(sys.argv) > 3 and sys.argv[3] == "long"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for i in range(n):
        ok, msg = case(rng, i, long_run)
        print(msg, flush=True)
        bad += not ok
    print("%d cases, %d failures, %.0f s" % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
