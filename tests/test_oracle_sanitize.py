"""The CPU oracle under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available
on the pool).  Runs in a subprocess because libasan must be preloaded."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
import hmmsort_amd as H
L = O.lib(%r)
O._LIB = L
temps = np.asfortranarray(np.stack([H.create_spike_template(12, 3.0, 0.8, 0.2),
                                    H.create_spike_template(12, 4.0, 0.3, 0.2)], 1))
pp = [0.02, 0.01]
for ov in (False, True):
    sm = O.state_matrix(2, 12, np.log(pp), ov)
    y = H.create_signal(300, 0.3, pp, temps, seed=5)
    x, ll = O.viterbi(y, sm, temps, 0.3, lean=True)
    x2, ll2, _ = O.viterbi(y, sm, temps, 0.3, return_T1=True)
    assert np.array_equal(x, x2) and ll == ll2
    mu = np.asfortranarray(temps * 0.9); mu[0, :] = 0
    O.train_step(y, sm, mu, 0.4)
    O.reconstruct_signal(x, sm, temps); O.unroll_mlseq(x, sm); O.extract_spiketimes(x, sm, temps)
    O.fit_chunked(y, sm, temps, 0.3, 100)
print("SANITIZED_OK")
'''


def test_oracle_under_asan_ubsan():
    so = os.path.join(ROOT, "oracle", "libhmm_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhmm_oracle_asan.so"],
                          stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", CODE % (ROOT, so)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "SANITIZED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
