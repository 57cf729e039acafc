"""Overlap decode of the reference's Viterbi-test model (N=2, K=60, allow_overlaps: 3600 states), device-resident:
the pair sweep (pair runs as delays) against the generic blocked sweep.  usage: python scripts/bench_pair.py [T] [mode]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
modes = sys.argv[2:] or ["pair", "generic"]
K = 60
temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2), H.create_spike_template(K, 4.0, 0.3, 0.2)], 1))
pp = [0.003, 0.001]
sm = H.StateMatrix.create(2, K, np.log(pp), True)
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
    t = time.perf_counter()
    for _ in range(3):
        plan.viterbi(dy, dx, dll, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    out[mode] = dx.cpu().numpy()
    print(mode, "%.2f ms  %.0f Msamples/s" % (dt * 1e3, T / dt / 1e6), plan.info(), plan.diagnostics(st))
    plan.close()
if len(out) == 2:
    print("same path:", np.array_equal(out["pair"], out["generic"]))
