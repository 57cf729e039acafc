"""Worker of tests/test_gpu_dist.py: two ranks on ONE GPU (gloo for the collectives; RCCL needs one GPU per
rank), through the product path: wave engine, batched plans, Plan.estep -> all-reduce -> Plan.mstep.

(1) pooled channels: every rank owns the channels shard_channels() deals it, sweeps them through one batched
    plan, sums its channels' statistics, ONE SUM all-reduce, M-step from the pooled vector;
(2) time shards of one recording: dist.time_shard_plan (certified shard edges), one SUM all-reduce.
Rank 0 also computes both in a single process and compares at 1e-9."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def allreduce_host(t):
    h = t.cpu()
    dist.all_reduce(h)
    t.copy_(h)
    return t


def main():
    dist.init_process_group(backend="gloo")   # before anything touches the GPU
    import hmmsort_amd as H
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    N, K, T, nch = 3, 40, 60_000, 4
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2),
                                        H.create_spike_template(K, 2.5, 0.6, 0.25)], 1))
    pp = [0.004, 0.002, 0.003]
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    mu = np.asfortranarray(temps * 0.9)
    mu[0, :] = 0
    ys = [H.create_signal(T, 0.3, pp, temps, seed=300 + ch) for ch in range(nch)]

    # ---- (1) pooled channels -------------------------------------------------------------------
    mine = H.dist.shard_channels(nch, rank, world)
    plan = H.Plan.batched(T, [sm] * len(mine), [mu] * len(mine), [0.35] * len(mine))
    assert plan.info()["engine"] == H.ENGINE_WAVE
    dy = torch.from_numpy(np.stack([ys[ch] for ch in mine])).cuda()
    stats = torch.zeros((len(mine), plan.stats_len()), dtype=torch.float64, device="cuda")
    plan.estep(dy, stats, st)
    d = plan.diagnostics(st)
    assert d[3] == 0 and d[5] == 0, d
    pooled = stats.sum(0)
    allreduce_host(pooled)
    out = torch.zeros((len(mine), plan.mstep_len()), dtype=torch.float64, device="cuda")
    stats[:] = pooled[None, :]
    plan.mstep(stats, out, st)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[0][:K * N + 1 + N], o[-1][:K * N + 1 + N])   # every channel gets the pooled model
    # all ranks hold the same pooled model
    mine_model = torch.from_numpy(o[0][:K * N + 1 + N].copy())
    ref_model = mine_model.clone()
    dist.broadcast(ref_model, 0)
    assert torch.equal(mine_model, ref_model)
    plan.close()
    if rank == 0:
        whole = H.Plan.batched(T, [sm] * nch, [mu] * nch, [0.35] * nch)
        dall = torch.from_numpy(np.stack(ys)).cuda()
        sall = torch.zeros((nch, whole.stats_len()), dtype=torch.float64, device="cuda")
        whole.estep(dall, sall, st)
        torch.cuda.synchronize()
        want, got = sall.sum(0).cpu().numpy(), pooled.cpu().numpy()
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12), np.abs(got - want).max()
        whole.close()

    # ---- (2) time shards of one recording --------------------------------------------------------
    Tl = 240_000
    yl = H.create_signal(Tl, 0.3, pp, temps, seed=77)
    plan, ysl, own = H.dist.time_shard_plan(yl, rank, world, sm, mu, 0.35, halo=512)
    part = torch.zeros(plan.stats_len(), dtype=torch.float64, device="cuda")
    plan.estep(torch.from_numpy(ysl).cuda(), part, st)
    d = plan.diagnostics(st)
    assert d[3] == 0 and d[5] == 0 and max(d[4], d[6]) < 1e-9, d    # incl. the boundaries inside the halos
    allreduce_host(part)
    out1 = torch.zeros(plan.mstep_len(), dtype=torch.float64, device="cuda")
    plan.mstep(part, out1, st)
    torch.cuda.synchronize()
    plan.close()
    if rank == 0:
        whole = H.Plan(Tl, sm, mu, 0.35)
        ref = torch.zeros(whole.stats_len(), dtype=torch.float64, device="cuda")
        whole.estep(torch.from_numpy(yl).cuda(), ref, st)
        out2 = torch.zeros(whole.mstep_len(), dtype=torch.float64, device="cuda")
        whole.mstep(ref, out2, st)
        torch.cuda.synchronize()
        r, t = ref.cpu().numpy(), part.cpu().numpy()
        assert np.allclose(t, r, rtol=1e-9, atol=1e-12), np.abs(t - r).max()
        a, b = out1.cpu().numpy()[:K * N + 1 + N], out2.cpu().numpy()[:K * N + 1 + N]
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
        whole.close()
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
