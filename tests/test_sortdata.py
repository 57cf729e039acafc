"""sort_data (reference src/hmmsort.jl:36-104) -- host logic on CPU, the full driver on the GPU."""
import numpy as np
import pytest

from conftest import to_oracle_sm


def test_get_lp_matches_closed_form(H):
    # types.jl:42-61: silent -> (a,1) carries lp[a] + (N-1)*lpz (types.jl:94-113), neuron order
    p = np.array([0.01, 0.02, 0.005])
    for ov in (True, False):
        sm = H.StateMatrix.create(3, 6, np.log(p), ov)
        lp, idx = H.get_lp(sm)
        lpz = np.log1p(-np.exp(np.log(p).sum()))
        assert np.array_equal(idx, [1, 2, 3])
        assert np.allclose(lp, np.log(p) + 2 * lpz, rtol=0, atol=1e-15)


def test_too_many_templates_bails_out(H):
    sf = np.zeros((10, 1, 5))
    assert H.sort_data(sf, [4.0], [0.01] * 5, np.zeros(100), dosave=False) == {}


@pytest.mark.gpu
def test_sort_data_end_to_end(O, H, tmp_path):
    from scipy.io import loadmat
    K, N, T = 24, 2, 250_000
    temps = np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                      H.create_spike_template(K, 4.0, 0.3, 0.2)], 1)
    temps[0, :] = 0.0
    p = np.array([0.004, 0.002])
    y = H.create_signal(T, 0.3, p, np.asfortranarray(temps), seed=4)
    raw = np.round(y * 1000).astype(np.int16)                 # an int16 recording, 2 channels
    data = np.stack([raw, raw[::-1]], 1)
    spike_forms = np.zeros((K, 3, N))
    spike_forms[:, 0, :] = temps * 1000
    sigma = 300.0
    src, dat, outp = tmp_path / "templates.npz", tmp_path / "data.npz", tmp_path / "out.mat"
    np.savez(src, spikeForms=spike_forms, cinv=np.array([[1.0 / sigma ** 2]]), p=p)
    np.savez(dat, data=data)
    from hmmsort_amd import sortdata
    assert sortdata.main(["--sourcefile", str(src), "--datafile", str(dat), "--outfile", str(outp)]) == 0
    out = loadmat(outp)
    # the same through the oracle: chunked decode with the reference's stitch rule, then unroll
    sm = H.StateMatrix.create(N, K, np.log(p), True)
    mu = np.asfortranarray(spike_forms[:, 0, :])
    rc, ml, ll = O.fit_chunked(raw.astype(np.float64), to_oracle_sm(O, sm), mu, sigma, 100_000)
    assert rc == 0
    assert np.array_equal(out["mlseq"], O.unroll_mlseq(ml, to_oracle_sm(O, sm)))
    assert abs(out["ll"].item() - ll) <= 1e-9 * abs(ll)
    assert np.allclose(out["waveforms"], mu) and abs(out["sigma"].item() - sigma) < 1e-9
    assert np.allclose(np.ravel(out["lp"]), H.get_lp(sm)[0])
    assert out["mlseq"].shape == (N, T) and out["mlseq"].max() > 1
