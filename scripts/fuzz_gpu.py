"""Randomised parity sweep: decode (all engines AUTO picks) and one EM step against the CPU oracle
over random model shapes, firing rates, noise levels and signal lengths.  Test infrastructure (it
imports the oracle): python scripts/fuzz_gpu.py [n_cases] [seed]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import hmmsort_amd as H
from oracle import oracle as O
from conftest import to_oracle_sm



def run(n_cases, seed, verbose=True):
    """returns the list of failing case descriptions"""
    rng = np.random.default_rng(seed)
    failures = []
    for case in range(n_cases):
        ok, tag, msg = one_case(rng, case)
        if not ok:
            failures.append(tag + " -> " + msg)
        if verbose:
            print(tag, "->", msg, flush=True)
    return failures


STRESS = bool(int(__import__("os").environ.get("FUZZ_STRESS", "0")))  # bigger, busier models
TIES = bool(int(__import__("os").environ.get("FUZZ_TIES", "0")))     # duplicate templates: tie-breaking rules
ENGINE = int(__import__("os").environ.get("FUZZ_ENGINE", "0"))        # 0 auto, 1 strict, 3 blocked (decode)
TIESCALE = bool(int(__import__("os").environ.get("FUZZ_TIESCALE", "0")))  # random multipliers of the wave engine's near-tie
#   threshold: ordinary decisions get flagged and the exact resolver (wave_ties.hip) has to re-decide them to the oracle's answer
PERSRC = bool(int(__import__("os").environ.get("FUZZ_PERSRC", "0")))      # exit->entry log-probabilities that depend on the source ring


def one_case(rng, case):
    if True:
        ov = bool(rng.integers(0, 2))
        N = int(rng.integers(1, 5 if ov else (13 if STRESS else 7)))
        K = int(rng.integers(2, 34 if ov else (200 if STRESS else 70)))
        smax = 12000 if STRESS else 6000
        if ov and 1 + N * (K - 1) + N * (N - 1) // 2 * (K - 1) ** 2 > smax:
            K = max(2, int(np.sqrt(smax / max(1, N * (N - 1) // 2))))
        T = int(rng.integers(300, 120000 if STRESS else 40000))
        sigma = float(rng.uniform(0.15, 0.6))
        temps = np.asfortranarray(np.stack([H.create_spike_template(K, rng.uniform(1.5, 5), rng.uniform(0.2, 1.0),
                                                                    rng.uniform(0.1, 0.4)) for _ in range(N)], 1))
        pp = rng.uniform(5e-4, 8e-3, N) * min(1.0, 30.0 / K) * (rng.choice([1.0, 3.0, 8.0]) if STRESS else 1.0)
        if TIES and N >= 2:   # identical templates with identical rates: every spike is an exact tie
            temps[:, 1] = temps[:, 0]; pp[1] = pp[0]
            if N >= 4:
                temps[:, 3] = temps[:, 2]; pp[3] = pp[2]
        y = H.create_signal(T, sigma, pp, temps, seed=int(rng.integers(1, 1 << 30)))
        sm = H.StateMatrix.create(N, K, np.log(pp), ov)
        if PERSRC and not ov and N >= 2 and rng.random() < 0.7:
            L_ = K - 1
            tr = sm.transitions.copy()
            for i_ in range(len(tr)):
                s_, d_ = int(tr["src"][i_]), int(tr["dst"][i_])
                if s_ > 1 and d_ > 1 and (s_ - 2) % L_ == L_ - 1 and (d_ - 2) % L_ == 0:
                    tr["lp"][i_] += rng.uniform(-0.5, 0.5)
            sm = H.StateMatrix(sm.states, tr, sm.pi, sm.K, sm.N, sm.nstates, False)
        osm = to_oracle_sm(O, sm)
        H.set_option("tie_scale", int(rng.choice([1, 1, 10 ** 6, 10 ** 8, 10 ** 9])) if TIESCALE else 1)
        blk = int(rng.choice([0, 0, 128, 192, 256, 512, 1024]))   # geometry requests: the engines clamp
        hal = int(rng.choice([0, 0, 64, 128, 256, 512]))          # them; a short warm-up must escalate
        H.set_option("block", blk)
        H.set_option("halo", hal)
        tag = "case %d: N=%d K=%d ov=%d S=%d T=%d sigma=%.2f block=%d halo=%d" % (case, N, K, ov, sm.nstates, T,
                                                                              sigma, blk, hal)
        try:
            H.set_option("engine", ENGINE)
            try:
                x, ll = H.viterbi(y, sm, temps, sigma)
            finally:
                H.set_option("engine", 0)
            esc = H.get_option("last_escalations")
            xo, llo = O.viterbi(y, osm, temps, sigma)
            ok = np.array_equal(x, xo) and abs(ll - llo) <= 1e-9 * abs(llo)
            msg = "viterbi %s (esc %d)" % ("ok" if ok else "MISMATCH %d samples, ll rel %.2e" % ((x != xo).sum(), abs(ll - llo) / abs(llo)), esc)
            if T >= 3000 and rng.random() < 0.5:   # chunked decode with the stitch rule, fit.jl:11-42
                cs = int(rng.integers(1000, T // 2))
                rc, ml, llc = O.fit_chunked(y, osm, temps, sigma, cs)
                try:
                    mdl = H.fit(H.HMMSpikeTemplateModel(sm, temps, sigma), y, cs)
                    ok3 = rc == 0 and np.array_equal(mdl.ml_seq, ml) and abs(mdl.ll - llc) <= 1e-9 * abs(llc)
                    if ok3:
                        sp = H.extract_spiketimes(mdl)
                        spo = O.extract_spiketimes(ml, osm, temps)
                        ok3 = all(np.array_equal(a_, b_) for a_, b_ in zip(sp, spo))
                except IndexError:
                    ok3 = rc == -3            # a chunk without a silent sample: both give up there
                msg += "; fit(chunk=%d) %s" % (cs, "ok" if ok3 else "MISMATCH rc=%d" % rc)
                ok = ok and ok3
            if T * sm.nstates <= 4_000_000 and T >= 2:
                mu = np.asfortranarray(temps * rng.uniform(0.8, 1.2, N)[None, :]); mu[0, :] = 0
                sm_n, mu_n, sig_n = H.train_step(y, sm, mu.copy(order="F"), sigma * 1.2)
                osm_n, omu, osig, olp, opp = O.train_step(y, osm, mu.copy(order="F"), sigma * 1.2)
                fin = np.isfinite(omu)
                ok2 = (np.array_equal(np.isfinite(mu_n), fin) and np.allclose(mu_n[fin], omu[fin], rtol=1e-6, atol=1e-9)
                       and (abs(sig_n - osig) <= 1e-6 * osig or not np.isfinite(osig)))
                msg += "; em_step %s" % ("ok" if ok2 else "MISMATCH max|dmu| %.2e dsig %.2e" % (np.nanmax(np.abs(mu_n - omu)), abs(sig_n - osig)))
                ok = ok and ok2
        except Exception as exc:  # noqa: BLE001
            ok, msg = False, "EXCEPTION %r" % (exc,)
        H.set_option("block", 0)
        H.set_option("halo", 0)
        H.set_option("tie_scale", 1)
    return ok, tag, msg


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    t00 = time.time()
    bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    print("%d cases, %d failures, %.0f s" % (n_cases, len(bad), time.time() - t00))
    sys.exit(1 if bad else 0)
