"""Rate of the reference CLI's decode: fit(HMMSpikingModel, templates, X, 100_000) (fit.jl:11-42,
hmmsort.jl:90) on an overlap-resolving model, host array in, host array out."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import hmmsort_amd as H

K = 60
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shapes = [(3.0, 0.8, 0.2), (4.0, 0.3, 0.2), (2.5, 0.6, 0.25), (3.5, 0.5, 0.15)]
temps = np.asfortranarray(np.stack([H.create_spike_template(K, *shapes[i]) for i in range(N)], 1))
pp = [0.003, 0.001, 0.002, 0.0015][:N]
sm = H.StateMatrix.create(N, K, np.log(pp), True)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
y = H.create_signal(T, 0.3, pp, temps, seed=3)
tm = H.HMMSpikeTemplateModel(sm, temps, 0.3)
for rep in range(2):
    t = time.time()
    m = H.fit(tm, y, 100_000)
    dt = time.time() - t
print("chunked fit, N=%d K=60 overlaps (%d states), %d samples in 100k chunks: %.3f s = %.1f Msamples/s, escalations %d"
      % (N, sm.nstates, T, dt, T / dt / 1e6, H.get_option("last_escalations")))
t = time.time(); m2 = H.fit(tm, y); dt = time.time() - t
print("whole-signal decode through hmmsort_viterbi: %.3f s = %.1f Msamples/s; same path as chunked: %s"
      % (dt, T / dt / 1e6, np.array_equal(m.ml_seq, m2.ml_seq)))
for rep in range(3):
    t = time.time(); x, ll = H.viterbi(y, sm, temps, 0.3); dt = time.time() - t
    print("hmmsort_viterbi call %d: %.3f s = %.1f Msamples/s (escalations %d)" % (rep, dt, T / dt / 1e6, H.get_option("last_escalations")))
