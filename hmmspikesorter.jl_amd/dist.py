"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm,
"gloo" in the CPU tests).

* channels are independent units in the reference (src/hmmsort.jl:79-83): `shard_channels` deals
  them round-robin; no collective on the data path;
* training ONE model on several shards (time shards of a recording, or pooled channels -- the
  latter is an extension the reference does not have) needs exactly one SUM all-reduce of the
  E-step statistics per EM iteration: `allreduce_stats`, between Plan.estep and Plan.mstep.
"""
import numpy as np


def shard_channels(n_channels, rank, world_size):
    """Channel indices owned by `rank` (round-robin)."""
    return list(range(rank, n_channels, world_size))


def time_shard(T, rank, world_size, halo):
    """Slice of a T-sample recording for `rank`: returns (slice_lo, slice_hi, own_lo, own_hi, first,
    last) with own_* in slice coordinates, for Plan.set_shard.  Owned ranges partition [0, T);
    every slice carries `halo` extra samples on each interior side (warm-up / ring completion)."""
    per = (T + world_size - 1) // world_size
    lo, hi = min(T, rank * per), min(T, (rank + 1) * per)
    first, last = rank == 0, hi >= T
    s_lo = 0 if first else max(0, lo - halo)
    s_hi = T if last else min(T, hi + halo)
    return s_lo, s_hi, lo - s_lo, hi - s_lo, first, last


def time_shard_plan(y, rank, world_size, lA, mu, sigma, halo=2048):
    """Plan + slice of recording `y` for `rank` with shard edges that carry a certificate: the wave engine
    certifies every chain boundary (a warm-up started from "silent, rings empty" against the neighbouring
    chain's own sweep), and a boundary inside each halo is what shows that the slice's arbitrary ends have
    been forgotten where the owned range begins (hmmsort_plan_set_shard refuses a shard without one).  The
    halo is widened until the plan's chain length fits into it.  Returns (plan, y_slice, (own_lo, own_hi))."""
    from . import _lib
    from .device import Plan
    T = len(y)
    for _ in range(8):
        s_lo, s_hi, o_lo, o_hi, first, last = time_shard(T, rank, world_size, halo)
        ys = np.ascontiguousarray(y[s_lo:s_hi])
        plan = Plan(len(ys), lA, mu, sigma)
        try:
            plan.set_shard(o_lo, o_hi, first, last)
            return plan, ys, (o_lo, o_hi)
        except _lib.HmmsortError:
            block = plan.info()["block"]
            plan.close()
            halo = max(2 * halo, 2 * block + 64)
    raise RuntimeError("time_shard_plan: no halo up to %d samples holds a chain boundary" % halo)


def allreduce_stats(stats, group=None):
    """In-place SUM all-reduce of a statistics vector (torch tensor on the rank's device)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def gather_results(local, n_channels, rank, world_size, group=None):
    """Collect per-channel python objects on every rank, in channel order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world_size == 1:
        return local
    parts = [None] * world_size
    dist.all_gather_object(parts, local, group=group)
    out = [None] * n_channels
    for r, items in enumerate(parts):
        for ch, item in zip(shard_channels(n_channels, r, world_size), items):
            out[ch] = item
    return out


def mstep_from_stats(stats, N, L):
    """Host (numpy) evaluation of the M-step finish from a statistics vector with the layout of
    hmmsort_plan_estep: [G0 | G1 | G2 | Xi | s_all | s_m | s_y2 | 0].  Only used to check the
    all-reduce plumbing in CPU tests; the product M-step is Plan.mstep on the GPU."""
    NL = N * L
    G0, G1, G2 = stats[:NL], stats[NL:2 * NL], stats[2 * NL:3 * NL]
    Xi = stats[3 * NL:3 * NL + N]
    s_all, s_m, s_y2 = stats[3 * NL + N:3 * NL + N + 3]
    mu = G1 / G0
    x2 = np.sum((G2 - 2 * mu * G1) + mu * mu * G0) + s_y2
    qq = np.sum(G0) + s_all
    mu_full = np.zeros((L + 1, N), order="F")
    mu_full[1:, :] = mu.reshape((N, L)).T
    return mu_full, float(np.sqrt(x2 / qq)), np.log(Xi) - np.log(s_m)
