"""HBM bandwidth probe on the GPU box: fill (write only), sum (read only), copy (read + write), in GB/s."""
import time
import torch
n = 400_000_000 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, f, nbytes in (("fill 400MB", lambda: x.fill_(1.0), n * 8), ("sum 400MB", lambda: x.sum(), n * 8),
                        ("copy 400MB", lambda: y.copy_(x), 2 * n * 8), ("memset", lambda: x.zero_(), n * 8)):
    dt = t(f)
    print("%-12s %.3f ms  %.0f GB/s" % (name, dt * 1e3, nbytes / dt / 1e9))
