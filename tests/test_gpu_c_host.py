"""The C ABI from a C host (tests/c_host/host.c, gcc, no Python in the process): state space helpers, decode
and one EM step with plain pointers must give what the ctypes binding gives for the same inputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_gets_the_same_answers(H, tmp_path):
    N, K, T = 3, 40, 150_000
    temps = np.asfortranarray(np.stack([H.create_spike_template(K, 2.5 + 0.7 * i, 0.3 + 0.2 * i, 0.2) for i in range(N)], 1))
    pp = np.array([0.003, 0.0015, 0.002])
    lp = np.log(pp)
    y = H.create_signal(T, 0.3, pp, temps, seed=12)
    mu0 = np.asfortranarray(temps * 0.9)
    mu0[0, :] = 0
    exe = str(tmp_path / "c_host")
    lib = os.path.join(ROOT, "hmmspikesorter.jl_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_host", "host.c"),
                           "-o", exe, "-L", lib, "-lhmmsort_hip", "-Wl,-rpath," + lib])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<qqq", N, K, T))
        f.write(struct.pack("<d", 0.35))
        f.write(lp.astype("<f8").tobytes())
        f.write(mu0.ravel(order="F").astype("<f8").tobytes())
        f.write(y.astype("<f8").tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(fout, "rb").read()
    S, nlp = struct.unpack_from("<qq", raw, 0)
    ll, sig = struct.unpack_from("<dd", raw, 16)
    off = 32
    x = np.frombuffer(raw, dtype="<i2", count=T, offset=off); off += 2 * T
    mu = np.frombuffer(raw, dtype="<f8", count=K * N, offset=off).reshape(N, K).T; off += 8 * K * N
    lp_new = np.frombuffer(raw, dtype="<f8", count=nlp, offset=off)
    # the same through the Python binding
    sm = H.StateMatrix.create(N, K, lp, False)
    assert S == sm.nstates and nlp == N
    x_py, ll_py = H.viterbi(y, sm, mu0, 0.35)
    sm_n, mu_py, sig_py = H.train_step(y, sm, mu0.copy(order="F"), 0.35)
    assert np.array_equal(x, x_py) and ll == ll_py
    assert np.array_equal(mu, mu_py) and sig == sig_py
    from hmmsort_amd.sortdata import get_lp
    # get_lp reads the entry probabilities back out of the rebuilt transition list (one rounding away)
    assert np.allclose(lp_new, get_lp(sm_n)[0], rtol=1e-8)
