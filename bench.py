#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): Msamples/s of (one Viterbi decode + one forward-backward E-step with its
sufficient statistics and M-step finish) over the same signal, K=4 templates x L=60 HMM
(reference naming: N=4 neurons, K=60 states per neuron, 237 states), 10 M samples per recording
channel, fp64, signal already resident in HBM.

One "step" = one pass of that pair of calls over one 10 M-sample channel.  With --gpus N every
rank (one process per GPU) owns one independent channel (weak scaling, no data-path collective:
channels are independent units in the reference); `value` is the whole-job
aggregate.  `--pooled` additionally SUM-all-reduces the E-step statistics over RCCL (templates
pooled across channels -- an extension the reference does not have; off by default).

Launch:  python bench.py --gpus 1 --steps 10 --warmup 3
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured streaming


def contract_bytes(S, T):
    """SURVEY.md section 8(d) contract figure: B(S) = 18*S + 28 bytes/sample for the combined metric (what a
    trellis-materialising implementation moves: forward writes alpha 8S, backward re-reads it, Viterbi
    writes psi 2S, ...).  The engines here never write the trellis, so this is NOT what they move; it is
    quoted as `frac_contract` only."""
    return (18 * S + 28) * T


def engine_bytes(kernel, N, T, info):
    """HBM bytes one launch of `kernel` has to move BY DESIGN (wave engine, natural-layout arrays; the
    delay lines live in LDS).  Used for the whole-step figure and as the fall-back when no rocprofv3
    counter summary of these very kernel sources is committed."""
    B, H = max(1, info["block"]), info["halo"]
    warm = 1.0 + H / B                       # every chain re-reads its warm-up
    pw = {True: 1, False: 2}[N <= 4] if N <= 8 else 4
    fused = N in (3, 4) and info.get("ring_len", 0) <= 64   # the backward sweep accumulates G1 itself (no kw_gsum, no rho)
    per_sample = {
        "kw_prepass": 8 + 8 * N + 8,                     # y in, N ring-score planes + the window sums W2 out
        "kw_vit": (8 + 8 * N) * warm + 4 * pw,           # y + ring scores in, packed back-pointers out
        "kw_fwd": (8 + 8 * N) * warm + 8 * (N + 2),      # ... la0, fref, N onset masses out
        "kw_bwd": (8 + 8 * N) * warm + 8 + 8 * (N + 2) + (0 if fused else 8 * N),   # y, ring scores, W2, forward outputs in; rho out
        "kw_gsum": 8 + 8 * N,                            # y, rho in
        "kw_backtrace": 4 * pw * 1.25 + 2,               # psi (with walk-in) in, x out
        "kw_ll_partial": 8 + 2,
        "kw_fb_check": 0.1, "kw_edges": 0.0,
    }.get(kernel)
    return None if per_sample is None else per_sample * T


def kernel_sources_hash():
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "hmmspikesorter.jl_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) +
                    glob.glob(os.path.join(src, "*.cpp"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(T, block, N=4, K=60, channels=1):
    """Per-kernel HBM bytes per launch from a committed rocprofv3 PMC summary (profiles/<tag>_summary.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 x2 read correction) -- but only
    of a summary taken with EXACTLY the kernel sources this process runs (_meta.kernel_sources) on the
    same workload (samples per channel, chain length, model shape, channels per plan); otherwise None, and
    the line says so."""
    import glob
    want = kernel_sources_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if (meta.get("kernel_sources") == want and meta.get("samples") == T and meta.get("block") == block and
                meta.get("neurons", 4) == N and meta.get("states", 60) == K and meta.get("channels", 1) == channels):
            measured_traffic.raw = d
            out = {k: v["hbm_read_bytes"] + v["hbm_write_bytes"] for k, v in d.items() if k != "_meta"}
            # the library's timing brackets name launch groups; rocprofv3 names kernels
            for alias, names in (("kw_gsum", ("kw_gsum_mx", "kw_gsum_generic")),
                                 ("kw_tie_prefix", ("kw_tie_bsum", "kw_tie_bscan", "kw_tie_btransfer")),
                                 ("kw_stitch_fix", ("kw_stitch_fix_par", "kw_stitch_fix"))):
                if alias not in out or alias == "kw_stitch_fix":
                    vals = [out[n] for n in names if n in out]
                    if vals:
                        out[alias] = sum(vals)
            for k, v in list(d.items()):
                if k != "_meta" and k in ("kw_gsum_mx",) and "kw_gsum" not in d:
                    d["kw_gsum"] = v
            return out, os.path.basename(path)
    return None, None


measured_traffic.raw = None


def shape_roofline(plan, info, N, K, T, channels, step_s, stream, profiled_step):
    """roofline block of one model shape: per-kernel HIP-event times of `profiled_step` (decode and E-step as
    separate calls, like the headline's profiling pass), the dominant kernel's HBM bytes (rocprofv3 counters of a
    committed summary of these very sources and this workload, else the bytes the kernel moves by design) over
    its launch time, and the same over the whole step."""
    plan.profile(True)
    profiled_step()
    prof = plan.profile_read(stream)
    plan.profile(False)
    ksum = {k: v[0] / v[1] for k, v in prof.items()}
    per_step = {k: v[0] for k, v in prof.items()}
    dom = max(per_step, key=per_step.get)
    counters, src = measured_traffic(T, info["block"], N, K, channels)
    model = {k: (engine_bytes(k, N, T, info) or 0.0) * channels for k in ksum}
    dom_counter = counters.get(dom) if counters else None
    dom_bytes = dom_counter if dom_counter is not None else model.get(dom)
    bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
    achieved = dom_bytes / (ksum[dom] * 1e-3) / 1e9 if dom_bytes else None
    if dom == "kw_gsum":   # Toeplitz product on the fp64 matrix cores: 2 flop x N L T useful flops (G1 only; G2 through W2)
        bound, peak, unit = "mfma", 78.6, "TFLOP/s"
        achieved = 2.0 * N * (K - 1) * T * channels / (ksum[dom] * 1e-3) / 1e12
    step_counters = sum(counters.get(k, 0.0) * prof[k][1] for k in ksum) if counters else None
    step_model = sum(model.values())
    return {"bound": bound, "kernel": dom, "achieved": achieved, "peak": peak, "unit": unit,
            "frac": (achieved / peak) if achieved else None,
            # fp64 v_mfma_f64_16x16x4 sustained on this box (scripts/micro/mfma_probe.hip): 36 TFLOP/s at one wave per
            # SIMD, 47 from two waves up -- below the spec figure used as `peak`
            "peak_measured": 47.0 if bound == "mfma" else None, "traffic": dom_counter,
            "traffic_source": src if dom_counter is not None else
            "none committed for these kernel sources and this workload: design bytes",
            "avg_launch_ms": ksum[dom],
            "kernel_ms": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            "step": {"ms": step_s * 1e3, "bytes_model": step_model, "bytes_counters": step_counters,
                     "frac": (step_counters or step_model) / step_s / 1e9 / HBM_PEAK_GBS}}


def bench_model(H, N, K):
    """the model/signal family of this benchmark (SURVEY 8d synthetic inputs scaled to N templates, K states)"""
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    pp = [[0.003, 0.001, 0.002, 0.0015][i % 4] * (60.0 / K) for i in range(N)]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    return temps, pp, H.StateMatrix.create(N, K, np.log(pp), False)


def run_config45(args, torch, H, dist, world, rank, dev):
    """BASELINE configs 4 and 5: many independent channels (the reference sorts one channel per call with its
    own model, src/hmmsort.jl:79-83), dealt round-robin over the ranks (dist.shard_channels); a rank sweeps its
    channels `batch` at a time through ONE batched plan.  No data-path collective; --pooled sums the
    statistics of all channels (one SUM all-reduce per step over RCCL) before the M-step -- pooled templates
    are an extension the reference lacks.  One step = decode + E-step + M-step finish of every channel once;
    value = all channels' samples / the slowest rank's time.  The total work is fixed: strong scaling."""
    cfg = args.config
    N, K = (8, 128) if cfg == 4 else (16, 256)
    if args.neurons != 4 or args.states != 60:
        N, K = args.neurons, args.states
    T = args.samples if args.samples != 10_000_000 or cfg == 4 else 100_000_000
    total = args.total_channels or (64 if cfg == 4 else 8)
    batch = args.batch or (8 if cfg == 4 else 1)
    sigma = 0.3
    temps, pp, sm = bench_model(H, N, K)
    S = sm.nstates
    mine = H.dist.shard_channels(total, rank, world)
    H.set_option("engine", args.engine)
    H.set_option("block", args.block)
    H.set_option("halo", args.halo)
    stream = torch.cuda.current_stream().cuda_stream
    groups = [mine[i:i + batch] for i in range(0, len(mine), batch)]
    plans = {}

    def plan_for(n):
        if n not in plans:
            plans[n] = H.Plan.batched(T, [sm] * n, [temps] * n, [sigma] * n)
        return plans[n]

    # signals stay resident in HBM (channel-major per group); per-channel seeds
    ys, xs, lls, sts, outs = [], [], [], [], []
    for grp in groups:
        yy = torch.empty((len(grp), T), dtype=torch.float64, device=dev)
        for i, ch in enumerate(grp):
            yy[i] = torch.from_numpy(H.create_signal(T, sigma, pp, temps, seed=1234 + ch)).to(dev)
        pl = plan_for(len(grp))
        ys.append(yy)
        xs.append(torch.zeros((len(grp), T), dtype=torch.int16, device=dev))
        lls.append(torch.zeros(len(grp), dtype=torch.float64, device=dev))
        sts.append(torch.zeros((len(grp), pl.stats_len()), dtype=torch.float64, device=dev))
        outs.append(torch.zeros((len(grp), pl.mstep_len()), dtype=torch.float64, device=dev))
    slen = plan_for(batch).stats_len() if groups else 0
    pooled_vec = torch.zeros(slen, dtype=torch.float64, device=dev)

    def step():
        for gi, grp in enumerate(groups):
            pl = plan_for(len(grp))
            pl.bind(ys[gi], stream)
            pl.decode_estep(ys[gi], xs[gi], lls[gi], sts[gi], stream)
            if not args.pooled:
                pl.mstep(sts[gi], outs[gi], stream)
            pl.unbind()
        if args.pooled:
            pooled_vec.zero_()
            for gi in range(len(groups)):
                pooled_vec.add_(sts[gi].sum(0))
            if dist is not None:
                if args.backend == "nccl":
                    dist.all_reduce(pooled_vec)
                else:
                    h = pooled_vec.cpu()
                    dist.all_reduce(h)
                    pooled_vec.copy_(h)
            for gi, grp in enumerate(groups):
                sts[gi][:] = pooled_vec[None, :]
                plan_for(len(grp)).mstep(sts[gi], outs[gi], stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # certificates and near-ties of every group, after the timed region
    bad = [0, 0, 0, 0]
    ties_done = 0
    for gi, grp in enumerate(groups):
        pl = plan_for(len(grp))
        pl.viterbi(ys[gi], xs[gi], lls[gi], stream)
        d1 = pl.diagnostics(stream)
        ties_done += pl.tie_stats(stream)["decided"]
        pl.estep(ys[gi], sts[gi], stream)
        d2 = pl.diagnostics(stream)
        bad = [bad[0] + d1[0], bad[1] + d2[3], bad[2] + d2[5], bad[3] + d1[7]]
    if dist is not None:
        bt = torch.tensor(bad, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(bt)
        bad = [int(v) for v in bt.tolist()]
    if rank == 0:
        roof, info = None, None
        if groups:
            pl = plan_for(len(groups[0]))
            info = pl.info()
            info["ring_len"] = K - 1
            gi = 0
            roof = shape_roofline(pl, info, N, K, T, len(groups[0]), dt / args.steps / max(1, len(groups)), stream,
                                  lambda: (pl.bind(ys[gi], stream), pl.viterbi(ys[gi], xs[gi], lls[gi], stream),
                                           pl.estep(ys[gi], sts[gi], stream), pl.mstep(sts[gi], outs[gi], stream),
                                           pl.unbind()))
        ms = dt / args.steps * 1e3
        res = {
            "metric": "Msamples/sec (Viterbi + forward-backward), K=%d L=%d HMM" % (N, K),
            "value": total * T / (dt / args.steps) / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %d channels x %d samples, K=%d L=%d HMM (reference N=%d, K=%d: %d "
                                   "states), channels dealt round-robin over the ranks, %d per batched plan: one Viterbi "
                                   "decode + one Baum-Welch E-step + M-step finish per channel, %s"
                                   % (cfg, total, T, N, K, N, K, S, batch,
                                      "statistics of all channels summed by ONE all-reduce per step (pooled templates: "
                                      "extension)" if args.pooled else "per-channel models (no collective)"),
                       "channels": total, "channels_per_rank": len(mine), "samples_per_channel": T, "states": S,
                       "engine": "wave", "block": info["block"] if info else None,
                       "halo": info["halo"] if info else None, "seed": 1234,
                       "pooled_allreduce": bool(args.pooled)},
            "roofline": roof,
            "detail": {"boundary_check_fails": bad[:3], "near_ties_unresolved": bad[3], "near_ties_resolved_rank0": ties_done,
                       "workspace_GB": info["workspace_bytes"] / 1e9 if info else None},
            "valid": bool(bad[0] == 0 and bad[1] == 0 and bad[2] == 0 and bad[3] == 0),
        }
        if world == 1 and not args.no_cpu_baseline and not args.quick:
            res["cpu_baseline"] = cpu_baseline(H, N, K, temps, pp, sigma, Tv=50_000 if cfg == 4 else 20_000,
                                               Tem=12_000 if cfg == 4 else 4_000, reps=1)
        print(json.dumps(res))
        if not res["valid"]:
            print("bench: certificates failed %s: the line above is INVALID" % (bad,), file=sys.stderr)
    for pl in plans.values():
        pl.close()


def cpu_baseline(H, N, K, temps, pp, sigma, Tv=100_000, Tem=40_000, reps=4):
    """The oracle (literal restatement of the reference loops) timed on a bounded sample of the same
    workload: Viterbi on 100 000-sample chunks (the reference's own chunk size, src/hmmsort.jl:90)
    and EM steps on 40 000-sample chunks (the reference materialises alpha/beta/gamma; both loops
    are O(T)).  The reference is single-threaded; SURVEY 8(d) also asks for one thread per chunk on
    all host cores, which is the figure reported as `value` (the single-thread rate is beside it)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    sm = O.state_matrix(N, K, np.log(pp), False)

    def work(seed, reps):
        y = H.create_signal(Tv, 0.3, pp, temps, seed=seed)
        t0 = time.perf_counter()
        for _ in range(reps):
            O.viterbi(y, sm, temps, sigma)
        t1 = time.perf_counter()
        for r in range(reps):
            lo = min(r * 15000, Tv - Tem)
            O.train_step(y[lo:lo + Tem], sm, np.asfortranarray(temps.copy()), sigma)
        t2 = time.perf_counter()
        return (t1 - t0) / (reps * Tv), (t2 - t1) / (reps * Tem)

    t_vit, t_em = work(99, 1)                      # one thread: the reference's execution model
    single = 1e-6 / (t_vit + t_em)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))                 # reps: ~4 s of work per thread at the headline shape (ctypes drops the GIL)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda i: work(200 + i, reps), range(cores)))
    wall = time.perf_counter() - t0
    # every thread did reps*(Tv decode + Tem EM) samples; rate of the combined metric = samples that
    # got BOTH a decode and an E-step per second, i.e. harmonic combination per thread
    per_thread = reps * (Tv * t_vit + Tem * t_em)  # single-thread time for the same work
    speedup = cores * per_thread / wall
    return {
        "value": single * speedup, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "single_thread_value": single,
        "sample": "oracle (C restatement, gcc -O2 -ffp-contract=off), one thread per chunk on %d host "
                  "cores: %d x (%d Viterbi decodes of %d samples + %d EM steps on %d samples) in "
                  "%.1f s = %.1fx the single thread (Viterbi %.2f, EM step %.4f Msamples/s single "
                  "thread); Julia is not installed, so the reference itself cannot be timed"
                  % (cores, cores, reps, Tv, reps, Tem, wall, speedup, 1e-6 / t_vit, 1e-6 / t_em),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=10_000_000, help="samples per channel")
    ap.add_argument("--pooled", action="store_true", help="all-reduce E-step statistics (extension)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-sharded", action="store_true",
                    help="strong scaling: ONE recording of --samples cut into time shards, one per "
                         "rank, statistics SUM-all-reduced every step (default: one channel per rank)")
    ap.add_argument("--channels", type=int, default=4,
                    help="extra (untimed) measurement: this many channels per GPU on concurrent streams")
    ap.add_argument("--separate", action="store_true",
                    help="decode and E-step as two calls instead of hmmsort_plan_decode_estep")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--neurons", type=int, default=4, help="templates (reference N); 4 = headline")
    ap.add_argument("--states", type=int, default=60, help="states per template (reference K)")
    ap.add_argument("--quick", action="store_true",
                    help="timed region and per-kernel profile only (no CPU baseline, EM loop, multi-channel, overlap extras): for rocprofv3 runs")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="BASELINE config: 2 (default; = 3's model) one 10 M-sample channel per GPU, K=4 L=60; "
                         "4: 64 channels x 10 M samples, K=8 L=128, dealt round-robin over the ranks, 8 channels per "
                         "batched plan, per-channel training (--pooled: ONE all-reduce of the channel-summed "
                         "statistics per step, an extension); 5: 8 channels x 100 M samples, K=16 L=256, one per rank")
    ap.add_argument("--total-channels", type=int, default=0, help="configs 4/5: channels of the whole job (default 64 / 8)")
    ap.add_argument("--batch", type=int, default=0, help="configs 4/5: channels per batched plan (default 8 / 1)")
    ap.add_argument("--engine", type=int, default=0, help="0 auto (wave), 2 lane-per-chain ring engine, 4 wave")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--halo", type=int, default=0)
    args = ap.parse_args()

    import torch
    import hmmsort_amd as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("HMMSORT_BENCH_ONE_GPU"):   # rehearsal: all ranks on GPU 0 (gloo only)
            local_rank = 0
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus, "launch one process per GPU (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a stream of our own: the library replays a call's launch sequence as one captured hipGraph, and capture
    # is not allowed on the legacy null stream (torch's default stream on ROCm)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    if args.config in (4, 5):
        run_config45(args, torch, H, dist, world, rank, dev)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- workload: BASELINE config 2/3 model ----
    N, K, T = args.neurons, args.states, args.samples
    base = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
    amps = [(base[i % 4][0] * (1 + 0.13 * (i // 4)), base[i % 4][1] + 0.03 * (i // 4), base[i % 4][2])
            for i in range(N)]
    pp = [[0.003, 0.001, 0.002, 0.0015][i % 4] * (60.0 / K) for i in range(N)]
    sigma = 0.3
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    seed = 1234 + (0 if args.time_sharded else rank)
    y = H.create_signal(T, sigma, pp, temps, seed=seed)
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    S = sm.nstates
    H.set_option("engine", args.engine)
    H.set_option("block", args.block)
    H.set_option("halo", args.halo)
    T_total = T
    if args.time_sharded and world > 1:
        # slice + halos; a certified chain boundary inside each halo (dist.time_shard_plan)
        plan, y, _ = H.dist.time_shard_plan(y, rank, world, sm, temps, sigma, halo=2048)
        T = len(y)
        args.pooled = True   # the shard statistics must be summed before the M-step
    else:
        plan = H.Plan(T, sm, temps, sigma)
    info = plan.info()
    info["ring_len"] = K - 1
    assert info["engine"] in (H.ENGINE_RING, H.ENGINE_WAVE)
    engine_name = {H.ENGINE_RING: "ring", H.ENGINE_WAVE: "wave"}[info["engine"]]

    stream = torch.cuda.current_stream().cuda_stream
    dy = torch.from_numpy(y).to(dev)
    dx = torch.zeros(T, dtype=torch.int16, device=dev)
    dll = torch.zeros(1, dtype=torch.float64, device=dev)
    stats = torch.zeros(plan.stats_len(), dtype=torch.float64, device=dev)
    out = torch.zeros(plan.mstep_len(), dtype=torch.float64, device=dev)

    def step():
        plan.bind(dy, stream)              # transpose + ring-score pre-pass, once per step
        if args.separate:
            plan.viterbi(dy, dx, dll, stream)
            plan.estep(dy, stats, stream)
        else:                              # same work, the three serial sweeps share one launch
            plan.decode_estep(dy, dx, dll, stats, stream)
        if args.pooled and dist is not None:
            if args.backend == "nccl":
                dist.all_reduce(stats)
            else:                                  # gloo rehearsal: reduce through the host
                h = stats.cpu()
                dist.all_reduce(h)
                stats.copy_(h)
        plan.mstep(stats, out, stream)
        plan.unbind()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- per-kernel timing (HIP events on the launch stream; separate, untimed pass) ----
    # Runs BEFORE the warm-up steps: the first ~15 ms of work on an idle GPU run below its sustained clocks
    # (10 timed steps after 3 warm-up steps: 1.25 ms/step, after 30: 1.18 ms/step), and this pass is untimed anyway.
    # decode and E-step as two calls here: in the timed step their kernels overlap on two streams, which
    # stretches every kernel's event-to-event time; run one after the other the durations are each kernel's own
    step()                      # loads every kernel's code object before anything is measured
    plan.viterbi(dy, dx, dll, stream)
    plan.estep(dy, stats, stream)
    fence()
    nprof = max(1, min(args.steps, 8))
    plan.profile(True)
    for _ in range(nprof):
        plan.bind(dy, stream)
        plan.viterbi(dy, dx, dll, stream)
        plan.estep(dy, stats, stream)
        plan.mstep(stats, out, stream)
        plan.unbind()
    prof = plan.profile_read(stream)
    plan.profile(False)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # boundary certificates: Viterbi calls fill diag[0..2], E-step calls diag[3..6]
    plan.viterbi(dy, dx, dll, stream)
    diag = plan.diagnostics(stream)
    plan.estep(dy, stats, stream)
    dE = plan.diagnostics(stream)
    diag = diag[:3] + dE[3:7] + diag[7:8]

    ksum = {k: v[0] / v[1] for k, v in prof.items()}           # average ms per launch
    per_step_ms = {k: v[0] / nprof for k, v in prof.items()}   # ms per step (a kernel may launch more than once)
    dom = max(ksum, key=ksum.get)
    step_ms_kernels = sum(per_step_ms.values())
    counters, counters_src = measured_traffic(T, info["block"], N, K) if info["engine"] == H.ENGINE_WAVE else (None, None)
    model = {k: engine_bytes(k, N, T, info) for k in ksum}
    dom_counter = counters.get(dom) if counters else None
    dom_bytes = dom_counter if dom_counter is not None else model.get(dom)
    bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
    achieved = dom_bytes / (ksum[dom] * 1e-3) / 1e9 if dom_bytes else None
    if dom == "kw_gsum":
        # the statistics kernel is a Toeplitz matrix product on the fp64 matrix cores: G1/G2[lag][ring] over
        # all onsets = 2 (G1,G2) x 2 flop x N L T useful flops; fp64 MFMA peak = the fp64 vector rate,
        # 78.6 TFLOP/s (MI355X_MICROARCH.md: FP32 vector 157.3 TF, fp64 at half rate)
        bound, peak, unit = "mfma", 78.6, "TFLOP/s"
        achieved = 4.0 * N * (K - 1) * T / (ksum[dom] * 1e-3) / 1e12
    # the fused backward sweep (3-4 rings, <= 64 states) is not bound by one pipe: besides its HBM streams it
    # issues the statistics product on the fp64 matrix cores and ~60 k vector instructions per wave
    other_roofs = None
    if dom == "kw_bwd" and info["engine"] == H.ENGINE_WAVE and N in (3, 4) and K - 1 <= 64:
        other_roofs = {"mfma_fp64": {"achieved": 2.0 * N * (K - 1) * T / (ksum[dom] * 1e-3) / 1e12, "peak": 78.6,
                                     "unit": "TFLOP/s (useful flops of G1)"}}
        raw = measured_traffic.raw
        if counters and raw and raw.get(dom, {}).get("valu_insts_per_wave"):
            lane_ops = raw[dom]["valu_insts_per_wave"] * 64.0 * info["nchains"]
            other_roofs["valu_fp64"] = {"achieved": lane_ops / ksum[dom] / 1e9, "peak": 35.0,
                                        "unit": "G lane-instructions per ms (peak measured: scripts/micro/fma_probe.hip)"}
        for v in other_roofs.values():
            v["frac"] = v["achieved"] / v["peak"]
    step_bytes_model = sum(v for v in model.values() if v)
    step_bytes_counters = (sum(counters.get(k, 0.0) * prof[k][1] / nprof for k in ksum) if counters else None)

    # split timings (untimed region): Viterbi only / E-step only
    def timed(fn, n=3):
        fence()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n
    plan.unbind()
    t_vit = timed(lambda: plan.viterbi(dy, dx, dll, stream))       # each incl. its own pre-pass
    t_est = timed(lambda: (plan.estep(dy, stats, stream), plan.mstep(stats, out, stream)))

    # ---- BASELINE config 3: full Baum-Welch EM, 10 iterations, device-resident (untimed extra) ----
    # every iteration = E-step + M-step on the GPU, then the host part the reference also has:
    # read back mu/sigma/lp (K*N+1+N doubles), rebuild the transition list, upload the new model
    def em_iterations(n_iter=10):
        rng = np.random.default_rng(7)
        sig0 = float(np.std(y, ddof=1))
        mu0 = np.ones((K, N), order="F")
        for i in range(N):  # the reference's random start, baumwelch.jl:311-322
            mu0[:, i] = H.create_spike_template(K, 3 * sig0 * rng.random(), 0.5 + 0.1 * rng.standard_normal(),
                                                1.5 * rng.random())
        mu0[0, :] = 0.0
        sm_i = H.StateMatrix.create(N, K, np.log(np.full(N, 2.0 ** (-3 * K / 2))), False)
        em = H.Plan(T, sm_i, mu0, sig0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n_iter):
            em.estep(dy, stats, stream)
            em.mstep(stats, out, stream)
            o = out.cpu().numpy()
            mu_i = np.asfortranarray(o[:K * N].reshape((K, N), order="F"))
            sm_i = H.StateMatrix.from_states(sm_i.states, o[K * N + 1 + N:], K, o[K * N + 1:K * N + 1 + N], False)
            em.set_model(sm_i, mu_i, float(o[K * N]))
        torch.cuda.synchronize()
        per = (time.perf_counter() - t) / n_iter
        em.close()
        return per * 1e3, float(o[K * N])
    try:
        em_ms, em_sigma = em_iterations() if not (args.time_sharded or args.quick) else (None, None)
    except H.HmmsortError as exc:
        # a template whose firing probability reaches 0 loses its entry transitions (the reference
        # keeps finite entries only, types.jl:121): the list length changes and the plan must be
        # rebuilt by the host, which this fixed-shape loop does not do
        em_ms, em_sigma = None, "stopped: %s" % exc

    # ---- several channels per GPU through ONE batched plan (hmmsort_plan_create_batched: chains =
    # channels x chains per channel, per-channel models; the shape of BASELINE config 4's per-GPU share;
    # untimed extra, reported in detail only) ----
    def multi_channel(nchan, Nc=N, Kc=K, Tc=T, steps=3):
        if (Nc, Kc) == (N, K):
            tm, ppc, smc = temps, pp, sm
        else:
            tm = np.asfortranarray(np.stack([H.create_spike_template(Kc, *amps_for(Nc)[i]) for i in range(Nc)], 1))
            ppc = [[0.003, 0.001, 0.002, 0.0015][i % 4] * (60.0 / Kc) for i in range(Nc)]
            smc = H.StateMatrix.create(Nc, Kc, np.log(ppc), False)
        pl = H.Plan.batched(Tc, [smc] * nchan, [tm] * nchan, [sigma] * nchan)
        yy = torch.empty((nchan, Tc), dtype=torch.float64, device=dev)
        for ch in range(nchan):
            yy[ch] = torch.from_numpy(H.create_signal(Tc, sigma, ppc, tm, seed=seed + 100 + ch)).to(dev)
        xx = torch.zeros((nchan, Tc), dtype=torch.int16, device=dev)
        ll_ = torch.zeros(nchan, dtype=torch.float64, device=dev)
        ss = torch.zeros((nchan, pl.stats_len()), dtype=torch.float64, device=dev)
        oo = torch.zeros((nchan, pl.mstep_len()), dtype=torch.float64, device=dev)
        def go():
            pl.bind(yy, stream); pl.decode_estep(yy, xx, ll_, ss, stream); pl.mstep(ss, oo, stream); pl.unbind()
        go(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            go()
        torch.cuda.synchronize()
        per = (time.perf_counter() - t) / steps
        dg = pl.diagnostics(stream)
        io = pl.info()
        io["ring_len"] = Kc - 1
        roof = shape_roofline(pl, io, Nc, Kc, Tc, nchan, per, stream,
                              lambda: (pl.bind(yy, stream), pl.viterbi(yy, xx, ll_, stream), pl.estep(yy, ss, stream),
                                       pl.mstep(ss, oo, stream), pl.unbind())) if (Nc, Kc) != (N, K) else None
        ties = pl.tie_stats(stream)
        pl.close()
        return {"channels_per_gpu": nchan, "model": "N=%d K=%d" % (Nc, Kc), "samples_per_channel": Tc,
                "Msamples_s": nchan * Tc / per / 1e6, "ms_per_step": per * 1e3, "block": io["block"],
                "chains": io["nchains"], "boundary_check_fails": [dg[0], dg[3], dg[5]], "near_ties_unresolved": dg[7],
                "near_ties_resolved": ties["decided"], "workspace_GB": io["workspace_bytes"] / 1e9, "roofline": roof}

    def amps_for(Nc):
        base_ = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
        return [(base_[i % 4][0] * (1 + 0.13 * (i // 4)), base_[i % 4][1] + 0.03 * (i // 4), base_[i % 4][2])
                for i in range(Nc)]
    extras = args.channels > 1 and not args.time_sharded and not args.quick and world == 1
    mc = multi_channel(args.channels) if extras else None
    # BASELINE config 4 (per-GPU share: 8 channels x 10 M, N=8, K=128) and config 5 (N=16, K=256; one
    # 40 M-sample channel here, the 100 M-sample run is tests/test_gpu_configs.py) -- untimed extras
    cfg4 = multi_channel(8, 8, 128, 10_000_000, steps=2) if (extras and (N, K) == (4, 60)) else None
    cfg5 = multi_channel(1, 16, 256, 40_000_000, steps=2) if (extras and (N, K) == (4, 60)) else None

    # ---- overlap-resolving decode: the reference's own Viterbi-test model,
    # test/runtests.jl:17-34 -- 2 templates, K=60, allow_overlaps=true, 3600 states -- through the
    # blocked engine, device-resident (untimed extra, rank 0 of a 1-GPU run only) ----
    def overlap_decode(To=10_000_000, No=2):
        Ko = 60
        shapes = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
        t2 = np.asfortranarray(np.stack([H.create_spike_template(Ko, *shapes[i]) for i in range(No)], 1))
        ppo = [0.003, 0.001, 0.002, 0.0015][:No]
        smo = H.StateMatrix.create(No, Ko, np.log(ppo), True)
        yo = torch.from_numpy(H.create_signal(To, sigma, ppo, t2, seed=seed + 7)).to(dev)
        xo = torch.zeros(To, dtype=torch.int16, device=dev)
        po = H.Plan(To, smo, t2, sigma)
        io = po.info()
        po.viterbi(yo, xo, dll, stream); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            po.viterbi(yo, xo, dll, stream)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t) / 3
        d = po.diagnostics(stream)
        po.close()
        sweep = ("pair sweep: pair runs as delays, lanes = phases" if No == 2 else
                 "multi sweep: one wavefront per template, pair runs as delays, compressed entry FIFOs")
        return {"model": "N=%d K=%d allow_overlaps=true, %d states" % (No, Ko, smo.nstates), "samples": To,
                "engine": {1: "strict", 2: "ring", 3: "blocked (%s)" % sweep}.get(io["engine"], io["engine"]),
                "block": io["block"], "halo": io["halo"], "Msamples_s": To / per / 1e6,
                "boundary_check_fails": d[0], "max_boundary_spread": d[2], "near_ties_on_path": d[7]}
    ov = overlap_decode() if (rank == 0 and world == 1 and not args.quick) else None
    # the largest model the reference's CLI builds (hmmsort.jl:50-54): 4 templates, 21 123 states
    ov4 = overlap_decode(10_000_000, 4) if (rank == 0 and world == 1 and not args.quick) else None

    ms = dt / args.steps * 1e3
    if rank == 0:
        res = {
            "metric": "Msamples/sec (Viterbi + forward-backward), K=%d L=%d HMM" % (N, K),
            "value": (T_total if args.time_sharded else world * T) / (dt / args.steps) / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong" if args.time_sharded else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "K=%d L=%d HMM (reference N=%d, K=%d: %d states), %d-sample single "
                                   "channel per GPU: one Viterbi decode + one Baum-Welch E-step "
                                   "(forward-backward + sufficient statistics + M-step finish)"
                                   % (N, K, N, K, S, T),
                       "channels": world, "samples_per_channel": T, "states": S,
                       "engine": engine_name, "block": info["block"], "halo": info["halo"],
                       "chains": info["nchains"], "seed": 1234, "pooled_allreduce": bool(args.pooled),
                       "time_sharded": bool(args.time_sharded)},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": peak,
                         "unit": unit, "frac": (achieved / peak) if achieved else None,
                         "traffic": dom_counter,
                         "traffic_source": counters_src if dom_counter is not None else
                         "none committed for these kernel sources: achieved/frac use the engine's design bytes",
                         "traffic_model": model.get(dom),
                         "avg_launch_ms": ksum[dom],
                         "frac_contract": contract_bytes(S, T) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "other_roofs": other_roofs,
                         "step": {"ms": ms, "bytes_model": step_bytes_model, "bytes_counters": step_bytes_counters,
                                  "frac": (step_bytes_counters or step_bytes_model) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                         "note": "frac = HBM bytes of the dominant kernel per launch (rocprofv3 FETCH_SIZE x2 + "
                                 "WRITE_SIZE of a committed profile of these exact kernel sources, else the bytes "
                                 "the kernel moves by design) / its HIP-event launch time / 8 TB/s; step.frac = "
                                 "the same over all kernels of one step and the wall time per step; "
                                 "frac_contract = SURVEY 8(d) trellis-materialising bytes (18 S + 28 per sample) "
                                 "/ step time / peak: > 1 because no engine here writes the trellis; other_roofs = "
                                 "the dominant kernel against the pipes it also loads (fused backward sweep)"},
            "detail": {"viterbi_Msamples_s": T / t_vit / 1e6, "estep_Msamples_s": T / t_est / 1e6,
                       "kernel_ms": {k: round(v, 4) for k, v in sorted(ksum.items(), key=lambda kv: -kv[1])},
                       "sum_kernel_ms_per_step": step_ms_kernels,
                       "em_iteration_ms": em_ms, "em_sigma_after_10": em_sigma,
                       "multi_channel": mc, "config4_share": cfg4, "config5_shape": cfg5,
                       "overlap_decode": ov, "overlap_decode_4_templates": ov4,
                       "diag": diag[:7], "workspace_GB": info["workspace_bytes"] / 1e9,
                       "near_ties": plan.tie_stats(stream)},
        }
        if world == 1 and not args.no_cpu_baseline and not args.quick:
            res["cpu_baseline"] = cpu_baseline(H, N, K, temps, pp, sigma)
        # a step whose chain boundaries are not all certified is not a measurement of the metric
        res["valid"] = bool(diag[0] == 0 and diag[3] == 0 and diag[5] == 0)
        res["detail"]["near_ties_on_path"] = diag[7] if len(diag) > 7 else None
        print(json.dumps(res))
        if not res["valid"]:
            print("bench: boundary certificates failed (diag %s): the line above is INVALID" % (diag,), file=sys.stderr)
            plan.close()
            sys.exit(3)
    plan.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
