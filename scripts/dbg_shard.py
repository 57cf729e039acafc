"""debug: time-sharded statistics vs the whole recording, entry by entry"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmmsort_amd as H  # noqa: E402
import torch  # noqa: E402

N, K = 3, 40
temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                    H.create_spike_template(K, 4.0, 0.3, 0.2),
                                    H.create_spike_template(K, 2.5, 0.6, 0.25)], 1))
pp = [0.004, 0.002, 0.003]
sm = H.StateMatrix.create(N, K, np.log(pp), False)
mu = np.asfortranarray(temps * 0.9)
mu[0, :] = 0
st = torch.cuda.current_stream().cuda_stream
L, NL = K - 1, N * (K - 1)
for Tl, world, halo in ((240_000, 2, 512), (90_000, 3, 256), (240_000, 2, 2048)):
    yl = H.create_signal(Tl, 0.3, pp, temps, seed=77)
    whole = H.Plan(Tl, sm, mu, 0.35)
    ref = torch.zeros(whole.stats_len(), dtype=torch.float64, device="cuda")
    whole.estep(torch.from_numpy(yl).cuda(), ref, st)
    total = torch.zeros_like(ref)
    for rank in range(world):
        plan, ys, own = H.dist.time_shard_plan(yl, rank, world, sm, mu, 0.35, halo=halo)
        part = torch.zeros_like(ref)
        dy = torch.from_numpy(ys).cuda()
        plan.estep(dy, part, st)
        d = plan.diagnostics(st)
        print("T=%d world=%d rank=%d slice=%d own=%s info=%s diag=%s" % (Tl, world, rank, len(ys), own, plan.info(), d[3:7]))
        total += part
        plan.close()
    r, t = ref.cpu().numpy(), total.cpu().numpy()
    bad = np.argsort(-np.abs(t - r))[:8]
    names = lambda i: ("G0[%d,%d]" % divmod(i, L) if i < NL else "G1[%d,%d]" % divmod(i - NL, L) if i < 2 * NL else
                       "G2[%d,%d]" % divmod(i - 2 * NL, L) if i < 3 * NL else "tail[%d]" % (i - 3 * NL))
    print("  max abs diff %.3e; worst:" % np.abs(t - r).max(), [(names(i), float(r[i]), float(t[i])) for i in bad])
    whole.close()
