"""Overlap decode with three or four templates (allow_overlaps; N=4, K=60: 21 123 states, the largest model the
reference's CLI builds), device-resident: the multi sweep (csrc/multi_sweep.hip) against the generic blocked sweep.
usage: python scripts/bench_multi.py [N] [K] [T] [modes...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
modes = sys.argv[4:] or ["multi", "generic"]
shapes = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15), (2.0, 0.4, 0.3)]
temps = np.asfortranarray(np.stack([H.create_spike_template(K, *shapes[i]) for i in range(N)], 1))
pp = [0.003, 0.001, 0.002, 0.0015, 0.001][:N]
sm = H.StateMatrix.create(N, K, np.log(pp), True)
y = H.create_signal(T, 0.3, pp, temps, seed=1241)
st = torch.cuda.current_stream().cuda_stream
dy = torch.from_numpy(y).cuda()
dll = torch.zeros(1, dtype=torch.float64, device="cuda")
out = {}
for mode in modes:
    if mode == "generic":
        os.environ["HMMSORT_PAIR"] = "0"
    else:
        os.environ.pop("HMMSORT_PAIR", None)
    plan = H.Plan(T, sm, temps, 0.3)
    dx = torch.zeros(T, dtype=torch.int16, device="cuda")
    plan.viterbi(dy, dx, dll, st)
    torch.cuda.synchronize()
    reps = 3 if mode != "generic" else 1
    t = time.perf_counter()
    for _ in range(reps):
        plan.viterbi(dy, dx, dll, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    out[mode] = dx.cpu().numpy()
    print(mode, "N=%d K=%d states %d: %.2f ms  %.0f Msamples/s" % (N, K, sm.nstates, dt * 1e3, T / dt / 1e6), plan.info(), plan.diagnostics(st), flush=True)
    plan.close()
if len(out) == 2:
    a, b = out[modes[0]], out[modes[1]]
    print("same path:", np.array_equal(a, b), "differing samples:", int((a != b).sum()))
