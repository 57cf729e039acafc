"""Cost of the exact near-tie resolver (wave_ties.hip): decode only, per-kernel HIP-event times.
usage: python scripts/bench_ties.py [N K T [tie_scale]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402


def model(N, K):
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    pp = [[0.003, 0.001, 0.002, 0.0015][i % 4] * (60.0 / K) for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    return temps, pp, H.StateMatrix.create(N, K, np.log(pp), False)


def main():
    N, K, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 256, 40_000_000)))
    scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    temps, pp, sm = model(N, K)
    y = H.create_signal(T, 0.3, pp, temps, seed=4321)
    H.set_option("tie_scale", scale)
    plan = H.Plan(T, sm, temps, 0.3)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).cuda()
    dx = torch.zeros(T, dtype=torch.int16, device="cuda")
    dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    for _ in range(2):
        plan.viterbi(dy, dx, dll, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.viterbi(dy, dx, dll, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    plan.profile(True)
    plan.viterbi(dy, dx, dll, st)
    prof = plan.profile_read(st)
    print("N=%d K=%d T=%d scale=%d: decode %.3f ms (%.0f Msamples/s)" % (N, K, T, scale, ms, T / ms / 1e3))
    print("  ties:", plan.tie_stats(st), "diag7:", plan.diagnostics(st)[7])
    for k, (m, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("  %-18s %8.3f ms x%d" % (k, m, n))
    plan.close()


if __name__ == "__main__":
    main()
