"""The reference's command-line driver, src/hmmsort.jl:36-104 (`sort_data`): templates + one
channel of raw data in, decoded state sequence and model out -- the callers either side of the hot
path.

The reference reads HDF5 (`spikeForms`, `cinv`, `p`; data under `rh/data/analogData` or
`highpassdata/data/data`) and writes a MAT file.  h5py is not available in this image, so the
inputs are arrays, `.npz` files or MATLAB v5 `.mat` files with the same variable names; the output
keys and shapes are the reference's (`mlseq`, `ll`, `waveforms`, `lp`, `sigma`, hmmsort.jl:94-98)
and are written with scipy.io.savemat.  Everything between load and save runs through the C ABI.
"""
import numpy as np

from . import api


def get_lp(lA):
    """get_lp(lA::StateMatrix) -> (lp, lidx)   types.jl:42-61: the log-probabilities of the
    transitions silent -> 'exactly one neuron active', in list order, with the neuron index."""
    lp = np.zeros(lA.N)
    lidx = np.zeros(lA.N, dtype=np.int64)
    k = 0
    for src, dst, val in lA.transitions:
        if src == 1 and dst > 1:
            vidx = np.nonzero(lA.states[:, dst - 1] > 1)[0]
            if len(vidx) == 1:
                lp[k] = val
                lidx[k] = vidx[0] + 1
                k += 1
                if k == lA.N:
                    break
    return lp, lidx


def _load(path, names):
    if str(path).endswith(".npz"):
        with np.load(path, allow_pickle=False) as f:
            return {n: f[n] for n in names if n in f}
    from scipy.io import loadmat
    f = loadmat(path)
    return {n: f[n] for n in names if n in f}


def load_templates(path):
    """spikeForms (nstates x nchannels x ntemplates), cinv, p   hmmsort.jl:39-48."""
    d = _load(path, ("spikeForms", "cinv", "p"))
    if "spikeForms" not in d:
        return None          # "No spike forms found. Bailing..."  hmmsort.jl:40-44
    return d["spikeForms"], np.atleast_1d(np.squeeze(d["cinv"])), np.atleast_1d(np.squeeze(d["p"]))


def load_data(path, name="data"):
    d = _load(path, (name,))
    return d[name]


def sort_data(spike_forms, cinv, p, data, outputfile=None, dosave=True, max_templates=4,
              chunksize=100_000):
    """sort_data(inputfile, datafile, outputfile; dosave, max_templates)   hmmsort.jl:36-104.

    Returns the reference's output dictionary; {} when there are more templates than
    `max_templates` (hmmsort.jl:49-52, 57-59)."""
    spike_forms = np.asarray(spike_forms, dtype=np.float64)
    nstates, _nchannels, ntemplates = spike_forms.shape
    pp = np.atleast_1d(np.asarray(p, dtype=np.float64))
    if len(pp) > max_templates:
        return {}
    # StateMatrix(ntemplates, nstates, log.(pp), true): the decode resolves overlaps  :53
    sm = api.StateMatrix.create(ntemplates, nstates, np.log(pp), True)
    sigma = float(np.sqrt(1.0 / np.atleast_1d(cinv)[0]))                      # sqrt(inv(cinv[1])) :55
    templates = api.HMMSpikeTemplateModel(sm, np.asfortranarray(spike_forms[:, 0, :]), sigma)
    lp, _ii = get_lp(sm)                                                       # :61
    data = np.asarray(data)
    if data.ndim == 2:
        data = data[:, 0]                                                      # view(data, :, 1) :80
    # :84-88 converts to Float64 on the host; int16 samples are handed over as they are and widened in HBM
    dataf = np.ascontiguousarray(data) if data.dtype == np.int16 else np.ascontiguousarray(data, dtype=np.float64)
    modelf = api.fit(templates, dataf, chunksize)                              # :90
    mlseq = api.unroll_mlseq(modelf.ml_seq, sm)                                # :92
    out = {"mlseq": mlseq, "ll": modelf.ll, "waveforms": templates.mu, "lp": lp, "sigma": sigma}
    if dosave and outputfile is not None:
        from scipy.io import savemat
        savemat(outputfile, out)                                               # MAT.matwrite :100
    return out


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="HMM spike sorting of one channel (hmmsort.jl CLI)")
    ap.add_argument("--sourcefile", required=True, help=".npz/.mat with spikeForms, cinv, p")
    ap.add_argument("--datafile", required=True, help=".npz/.mat with the channel's samples")
    ap.add_argument("--dataname", default="data")
    ap.add_argument("--outfile", default="hmmsort.mat")
    ap.add_argument("--max_templates", type=int, default=4)
    a = ap.parse_args(argv)
    t = load_templates(a.sourcefile)
    if t is None:
        print("No spike forms found. Bailing...")
        return 0
    out = sort_data(*t, load_data(a.datafile, a.dataname), a.outfile, max_templates=a.max_templates)
    print("Done! Results saved to %s" % a.outfile if out else "Too many templates. Bailing out...")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
