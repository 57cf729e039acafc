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


def algorithmic_bytes(kernel, S, T):
    """B(S) = 18*S + 28 bytes/sample for the combined metric, split by
    sweep: forward reads y (8) and writes alpha (8S); backward re-reads y and alpha (8S+8);
    Viterbi reads y (8), writes psi as Int16 per state (2S); backtrace reads >= 2 and writes x (2)."""
    per_sample = {
        "k_vfb_chain": (2 * S + 8) + 2 * (8 * S + 8),  # Viterbi + forward + backward sweeps, one launch
        "k_fb_chain": 2 * (8 * S + 8),      # forward and backward sweeps in one launch
        "k_fwd_chain": 8 * S + 8,
        "k_bwd_chain": 8 * S + 8,
        "k_vit_chain": 2 * S + 8,
        "k_vit_backtrace": 4,
    }.get(kernel, 8)
    return per_sample * T


def measured_traffic(kernel, T, block, halo):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same
    command (profiles/<tag>_summary.json, written by profiles/summarize.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2 read correction).  None when no
    summary matches the workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("samples") == T and meta.get("block") == block and meta.get("halo") == halo \
                and kernel in d:
            best = (d[kernel]["hbm_read_bytes"] + d[kernel]["hbm_write_bytes"], os.path.basename(path))
    return best


def cpu_baseline(H, N, K, temps, pp, sigma):
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
    Tv, Tem = 100_000, 40_000

    def work(seed, reps):
        y = H.create_signal(Tv, 0.3, pp, temps, seed=seed)
        t0 = time.perf_counter()
        for _ in range(reps):
            O.viterbi(y, sm, temps, sigma)
        t1 = time.perf_counter()
        for r in range(reps):
            O.train_step(y[r * 15000:r * 15000 + Tem], sm, np.asfortranarray(temps.copy()), sigma)
        t2 = time.perf_counter()
        return (t1 - t0) / (reps * Tv), (t2 - t1) / (reps * Tem)

    t_vit, t_em = work(99, 1)                      # one thread: the reference's execution model
    single = 1e-6 / (t_vit + t_em)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    reps = 4                                       # ~4 s of work per thread (ctypes drops the GIL)
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
                  "cores: %d x (%d Viterbi decodes of 100k samples + %d EM steps on 40k samples) in "
                  "%.1f s = %.1fx the single thread (Viterbi %.2f, EM step %.4f Msamples/s single "
                  "thread); Julia is not installed, so the reference itself cannot be timed"
                  % (cores, cores, reps, reps, wall, speedup, 1e-6 / t_vit, 1e-6 / t_em),
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
        s_lo, s_hi, o_lo, o_hi, first, last = H.dist.time_shard(T, rank, world, halo=2048)
        y = np.ascontiguousarray(y[s_lo:s_hi])
        T = len(y)
    plan = H.Plan(T, sm, temps, sigma)
    if args.time_sharded and world > 1:
        plan.set_shard(o_lo, o_hi, first, last)
        args.pooled = True   # the shard statistics must be summed before the M-step
    info = plan.info()
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
    diag = diag[:3] + dE[3:7]

    # ---- per-kernel timing (HIP events on the launch stream; separate, untimed pass) ----
    plan.profile(True)
    for _ in range(max(1, min(args.steps, 5))):
        step()
    prof = plan.profile_read(stream)
    plan.profile(False)
    ksum = {k: v[0] / v[1] for k, v in prof.items()}           # average ms per launch
    dom = max(ksum, key=ksum.get)
    achieved = algorithmic_bytes(dom, S, T) / (ksum[dom] * 1e-3) / 1e9
    step_ms_kernels = sum(v[0] for v in prof.values()) / max(1, min(args.steps, 5))
    traffic = measured_traffic(dom, T, info["block"], info["halo"])

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

    # ---- several channels per GPU, one plan + one stream each (the serving shape of a multi-
    # channel probe; untimed extra, reported in detail only) ----
    def multi_channel(nchan):
        plans, streams, bufs = [], [], []
        for ch in range(nchan):
            pl = H.Plan(T, sm, temps, sigma)
            stc = torch.cuda.Stream()
            yy = dy if ch == 0 else torch.from_numpy(H.create_signal(T, sigma, pp, temps, seed=seed + 100 + ch)).to(dev)
            bufs.append((yy, torch.zeros(T, dtype=torch.int16, device=dev),
                         torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros_like(stats),
                         torch.zeros_like(out)))
            plans.append(pl); streams.append(stc)
        def go():
            for pl, stc, (yy, xx, ll_, ss, oo) in zip(plans, streams, bufs):
                h = stc.cuda_stream
                pl.bind(yy, h); pl.decode_estep(yy, xx, ll_, ss, h); pl.mstep(ss, oo, h); pl.unbind()
        go(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        per = (time.perf_counter() - t) / 3
        for pl in plans:
            pl.close()
        return nchan * T / per / 1e6
    mc = multi_channel(args.channels) if (args.channels > 1 and not args.time_sharded and not args.quick) else None

    # ---- overlap-resolving decode: the reference's own Viterbi-test model,
    # test/runtests.jl:17-34 -- 2 templates, K=60, allow_overlaps=true, 3600 states -- through the
    # blocked engine, device-resident (untimed extra, rank 0 of a 1-GPU run only) ----
    def overlap_decode(To=2_000_000):
        Ko = 60
        t2 = np.asfortranarray(np.stack([H.create_spike_template(Ko, 3.0, 0.8, 0.2),
                                         H.create_spike_template(Ko, 4.0, 0.3, 0.2)], 1))
        ppo = [0.003, 0.001]
        smo = H.StateMatrix.create(2, Ko, np.log(ppo), True)
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
        return {"model": "N=2 K=%d allow_overlaps=true, %d states" % (Ko, smo.nstates), "samples": To,
                "engine": {1: "strict", 2: "ring", 3: "blocked"}.get(io["engine"], io["engine"]),
                "block": io["block"], "halo": io["halo"], "Msamples_s": To / per / 1e6,
                "boundary_check_fails": d[0], "max_boundary_spread": d[2], "near_tie_blocks": d[7]}
    ov = overlap_decode() if (rank == 0 and world == 1 and not args.quick) else None

    if rank == 0:
        ms = dt / args.steps * 1e3
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
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "traffic_GBps": (traffic[0] / (ksum[dom] * 1e-3) / 1e9) if traffic else None,
                         "avg_launch_ms": ksum[dom],
                         "note": "achieved = SURVEY 8(d) algorithmic bytes of the sweep this kernel "
                                 "implements (a trellis-materialising sweep) / HIP-event launch time; "
                                 "the ring engine never writes the trellis (junction-only recursion), "
                                 "so frac > 1; traffic = HBM bytes per launch actually moved "
                                 "(rocprofv3 PMC), traffic_GBps = that over the same launch time"},
            "detail": {"viterbi_Msamples_s": T / t_vit / 1e6, "estep_Msamples_s": T / t_est / 1e6,
                       "kernel_ms": {k: round(v, 4) for k, v in sorted(ksum.items(), key=lambda kv: -kv[1])},
                       "sum_kernel_ms_per_step": step_ms_kernels,
                       "em_iteration_ms": em_ms, "em_sigma_after_10": em_sigma,
                       "multi_channel": {"channels_per_gpu": args.channels, "Msamples_s": mc} if mc else None,
                       "overlap_decode": ov,
                       "diag": diag[:7], "workspace_GB": info["workspace_bytes"] / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline and not args.quick:
            res["cpu_baseline"] = cpu_baseline(H, N, K, temps, pp, sigma)
        print(json.dumps(res))
    plan.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
