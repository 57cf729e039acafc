"""The host-buffer entry points (hmmsort_viterbi / hmmsort_em_step: what a reference-side binding calls,
INTEGRATION.md) keep their plan and device buffers between calls: results must not depend on that, two host
threads must be able to decode at once, and hmmsort_shutdown must give the memory back."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def model(H, N=4, K=60):
    amps = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)][:N]
    pp = [0.003, 0.001, 0.002, 0.0015][:N]
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, *a) for a in amps], 1))
    return temps, pp, H.StateMatrix.create(N, K, np.log(pp), False)


def test_cached_plan_gives_the_same_answers_as_a_fresh_one(H):
    temps, pp, sm = model(H)
    T = 300_000
    ys = [H.create_signal(T, 0.3, pp, temps, seed=s) for s in (1, 2)]
    mus = [temps, np.asfortranarray(temps * 0.9)]
    H.set_option("plan_cache", 0)
    ref = [(H.viterbi(y, sm, mu, sg), H.train_step(y, sm, mu.copy(order="F"), sg))
           for y, mu, sg in zip(ys, mus, (0.3, 0.35))]
    H.set_option("plan_cache", 4)
    try:
        for rep in range(2):               # second round runs on plans left by the first
            for (y, mu, sg), ((x0, ll0), (sm0, mu0, s0)) in zip(zip(ys, mus, (0.3, 0.35)), ref):
                x, ll = H.viterbi(y, sm, mu, sg)
                assert np.array_equal(x, x0) and ll == ll0
                sm1, mu1, s1 = H.train_step(y, sm, mu.copy(order="F"), sg)
                assert np.array_equal(mu1, mu0) and s1 == s0
                assert np.array_equal(sm1.transitions["lp"], sm0.transitions["lp"])
    finally:
        H.shutdown()


def test_two_host_threads_decode_concurrently(H):
    temps, pp, sm = model(H)
    T = 2_000_000
    ys = [H.create_signal(T, 0.3, pp, temps, seed=10 + i) for i in range(4)]
    expect = [H.viterbi(y, sm, temps, 0.3) for y in ys]
    out = [None] * 4
    errs = []

    def work(i):
        try:
            for _ in range(3):
                out[i] = H.viterbi(ys[i], sm, temps, 0.3)
                r = H.train_step(ys[i], sm, temps.copy(order="F"), 0.3)
                assert np.isfinite(r[2])
        except Exception as e:                  # noqa: BLE001 - reported below
            errs.append(e)

    # options are read and written while the workers run
    stop = threading.Event()

    def poke():
        while not stop.is_set():
            H.set_option("escalate", 1)
            H.get_option("engine")

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    pk = threading.Thread(target=poke)
    pk.start()
    [t.start() for t in th]
    [t.join() for t in th]
    stop.set()
    pk.join()
    H.shutdown()
    assert not errs, errs
    for (x, ll), (x0, ll0) in zip(out, expect):
        assert np.array_equal(x, x0) and ll == ll0


def test_two_host_threads_overlap_on_the_device(H):
    # every host-buffer call works on its slot's own stream and waits for that stream only (no
    # hipDeviceSynchronize): two threads decoding 10 M samples each take clearly less than twice one thread.
    # What cannot overlap is the PCIe link both share: a float64 recording moves 100 MB per call (80 up, 20
    # down: 1.75 of the 2.9 ms), the acquisition's own int16 samples (src/hmmsort.jl:79-88) 40 MB.
    import time
    temps, pp, sm = model(H)
    T = 10_000_000
    ys = [H.create_signal(T, 0.3, pp, temps, seed=70 + i) for i in range(2)]
    raws = [np.round(y * 1000.0).astype(np.int16) for y in ys]       # microvolt-like integer samples
    cases = {"float64": (ys, temps, 0.3, 1.75), "int16": (raws, np.asfortranarray(temps * 1000.0), 300.0, 1.5)}
    for name, (sig, tm, sg, bound) in cases.items():
        for y in sig:                      # warm: plans and buffers of both slots exist
            H.viterbi(y, sm, tm, sg)
        th = [threading.Thread(target=H.viterbi, args=(sig[i], sm, tm, sg)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]             # now two slots are cached

        def one(n):
            t0 = time.perf_counter()
            for _ in range(n):
                H.viterbi(sig[0], sm, tm, sg)
            return (time.perf_counter() - t0) / n

        def two(n):
            def w(i):
                for _ in range(n):
                    H.viterbi(sig[i], sm, tm, sg)
            t0 = time.perf_counter()
            tt = [threading.Thread(target=w, args=(i,)) for i in range(2)]
            [t.start() for t in tt]
            [t.join() for t in tt]
            return (time.perf_counter() - t0) / n

        t1 = min(one(5), one(5))
        t2 = min(two(5), two(5))
        print("host-buffer decode of 10 M %s samples: one thread %.2f ms, two threads %.2f ms per round (%.2fx)"
              % (name, t1 * 1e3, t2 * 1e3, t2 / t1))
        H.shutdown()
        assert t2 < bound * t1, (name, t1, t2)


def test_shutdown_frees_the_cached_buffers_and_em_step_is_cheap_when_cached(H):
    import torch
    temps, pp, sm = model(H)
    T = 10_000_000
    y = H.create_signal(T, 0.3, pp, temps, seed=5)
    H.shutdown()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    H.train_step(y, sm, temps.copy(order="F"), 0.3)            # builds the plan
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 > T * 8                                # signal + workspace stay resident
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        H.train_step(y, sm, temps.copy(order="F"), 0.3)
        ts.append(time.perf_counter() - t0)
    H.set_option("plan_cache", 0)
    t0 = time.perf_counter()
    H.train_step(y, sm, temps.copy(order="F"), 0.3)
    cold = time.perf_counter() - t0
    H.set_option("plan_cache", 4)
    print("hmmsort_em_step, 10 M samples from a pageable host buffer: cached %.2f ms, uncached %.2f ms"
          % (1e3 * min(ts), 1e3 * cold))
    assert min(ts) < cold
    H.shutdown()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)


def test_int16_samples_are_widened_on_the_device(H):
    # hmmsort.jl:79-88: column 1 of the acquisition array, converted to Float64, goes into fit
    import torch
    K = 24
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                        H.create_spike_template(K, 4.0, 0.3, 0.2)], 1)) * 1000
    pp = [0.004, 0.002]
    sm = H.StateMatrix.create(2, K, np.log(pp), False)
    T = 400_000
    raw = np.round(H.create_signal(T, 300.0, pp, temps, seed=8)).astype(np.int16)
    x0, ll0 = H.viterbi(raw.astype(np.float64), sm, temps, 300.0)
    x1, ll1 = H.viterbi(raw, sm, temps, 300.0)                     # hmmsort_viterbi_i16
    assert np.array_equal(x0, x1) and ll0 == ll1 and x0.max() > 1
    tm = H.HMMSpikeTemplateModel(sm, temps, 300.0)
    f0 = H.fit(tm, raw.astype(np.float64), 100_000)
    f1 = H.fit(tm, raw, 100_000)
    assert np.array_equal(f0.ml_seq, f1.ml_seq) and f0.ll == f1.ll
    # a sample-major recording of 3 channels in device memory: channel 1 with stride 3
    rec = np.stack([raw, raw[::-1], -raw], 1)
    d_rec = torch.from_numpy(rec).cuda()
    for dt, code in ((torch.int16, 0), (torch.int32, 1), (torch.float32, 2), (torch.float64, 3)):
        src = d_rec.to(dt).contiguous()
        out = torch.empty(T, dtype=torch.float64, device="cuda")
        H._lib.check(H._lib.lib().hmmsort_samples_to_f64(src.data_ptr() + src.element_size(), code, T, 3,
                                                          out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), rec[:, 1].astype(np.float64))
    H.shutdown()


def test_strict_fallback_that_does_not_fit_returns_the_time_parallel_path(H):
    # duplicate templates: every spike is a tie the reference breaks by its own rounding.  The wave engine
    # flags those decisions and re-decides them with the reference's serial arithmetic (wave_ties.hip): the
    # path is the strict engine's without any escalation.  With the resolver switched off (test aid
    # tie_debug = 2: every flagged decision stays open, diag[7] > 0) hmmsort_viterbi re-decodes with the strict
    # engine -- unless the strict sweep's S x T back-pointer table does not fit (here: a 1 MB limit); then the
    # time-parallel path is returned, last_escalations < 0 says how many decisions are open, and the path is
    # as likely as the strict one (equal ll).
    import warnings
    K, N, T = 40, 2, 200_000
    t1 = H.create_spike_template(K, 3.0, 0.8, 0.2)
    temps = np.asfortranarray(np.stack([t1, t1], 1))
    pp = [0.004, 0.004]
    sm = H.StateMatrix.create(N, K, np.log(pp), False)
    y = H.create_signal(T, 0.3, pp, temps, seed=3)
    H.set_option("engine", H.ENGINE_STRICT)
    xs, lls = H.viterbi(y, sm, temps, 0.3)
    H.set_option("engine", H.ENGINE_AUTO)
    H.shutdown()
    xr, llr = H.viterbi(y, sm, temps, 0.3)                       # AUTO: wave engine + exact resolver
    assert H.get_option("last_escalations") == 0
    assert np.array_equal(xr, xs) and abs(llr - lls) <= 1e-9 * abs(lls)
    H.shutdown()
    H.set_option("tie_debug", 2)
    try:
        x0, ll0 = H.viterbi(y, sm, temps, 0.3)                   # resolver off: ties -> strict engine
        assert H.get_option("last_escalations") >= 1
        assert np.array_equal(x0, xs)
        H.set_option("strict_limit_mb", 1)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            x1, ll1 = H.viterbi(y, sm, temps, 0.3)
        open_ties = -H.get_option("last_escalations")
        assert open_ties >= 1 and len(w) == 1 and "near-tie" in str(w[0].message)
    finally:
        H.set_option("strict_limit_mb", 0)
        H.set_option("tie_debug", 0)
        H.shutdown()
    assert abs(ll1 - ll0) <= 1e-9 * abs(ll0)
    # the two paths agree except for which of the twin rings carries a spike
    assert np.array_equal(x0 > 1, x1 > 1)


def test_cached_slot_is_resized_when_the_list_changes(O, H):
    # the slot key holds neither R nor the engine.  Two overlap models with the same state table but a
    # different number of finite entry log-probabilities (the reference drops -Inf transitions, types.jl:121) need
    # statistics / output buffers of different lengths.  Every step must equal the step of a fresh plan, whatever
    # the cache held before.  (hmmsort_em_step is called directly: rebuilding a StateMatrix from the dead model's
    # shortened lp fails in the reference too -- isvalid_transition indexes lp[2].)
    import ctypes as C
    from conftest import to_oracle_sm
    from hmmsort_amd._lib import check, lib, ptr, TRANS_DTYPE
    K, T = 20, 30_000
    t1 = H.create_spike_template(K, 3.0, 0.8, 0.2)
    t2 = H.create_spike_template(K, 4.0, 0.3, 0.2)
    temps = np.asfortranarray(np.stack([t1, t2], 1))
    y = H.create_signal(T, 0.3, [0.004, 0.002], temps, seed=12)
    full = H.StateMatrix.create(2, K, np.log([0.004, 0.002]), True)
    dead = H.StateMatrix.create(2, K, np.array([np.log(0.004), -np.inf]), True)   # neuron 2 cannot start
    assert len(dead.transitions) < len(full.transitions)

    def em_step(sm):
        st = np.asfortranarray(sm.states, dtype=np.int16)
        tr = np.ascontiguousarray(sm.transitions, dtype=TRANS_DTYPE)
        mu = temps.copy(order="F")
        sig, nlp = C.c_double(0.0), C.c_int64(0)
        lp, pp = np.zeros(len(tr)), np.zeros(sm.nstates)
        check(lib().hmmsort_em_step(ptr(y), len(y), ptr(st), sm.N, sm.K, sm.nstates, ptr(tr), len(tr), ptr(mu), 0.35,
                                    C.cast(C.byref(sig), C.c_void_p), ptr(lp), len(lp), C.byref(nlp), ptr(pp)))
        return mu, sig.value, lp[:nlp.value].copy(), pp

    models = [full, dead, full, dead]
    H.set_option("plan_cache", 0)
    ref = [em_step(sm) for sm in models[:2]]
    assert len(ref[0][2]) == 3 and len(ref[1][2]) == 1        # xb[2:end]: one entry per transition leaving state 1 but the first
    H.set_option("plan_cache", 4)
    try:
        for i, sm in enumerate(models):
            mu1, s1, lp1, pp1 = em_step(sm)
            mu0, s0, lp0, pp0 = ref[i % 2]
            fin = np.isfinite(mu0)
            # (the dead model's unreachable states turn the reference's update() into NaN: logsumexpl(-Inf, -Inf);
            # what is compared is cached against fresh, bit for bit, NaN for NaN)
            assert np.array_equal(np.isfinite(mu1), fin) and np.array_equal(mu1[fin], mu0[fin]), i
            assert s1 == s0 or (np.isnan(s1) and np.isnan(s0)), i
            assert np.array_equal(lp1, lp0, equal_nan=True) and np.array_equal(pp1, pp0, equal_nan=True)
        # and the live model's step is the oracle's
        _, omu, osig, _, _ = O.train_step(y, to_oracle_sm(O, full), temps.copy(order="F"), 0.35)
        assert np.allclose(ref[0][0], omu, rtol=1e-6, atol=1e-9) and abs(ref[0][1] - osig) <= 1e-6 * osig
    finally:
        H.shutdown()
