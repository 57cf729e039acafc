"""Device-resident plan API (hmmsort_plan_* in include/hmmsort.h) for hosts that keep the signal
in HBM: bench.py and the multi-GPU driver.  Buffers are plain device pointers; torch is used only
as the allocator/stream provider (tensor.data_ptr(), torch.cuda.current_stream().cuda_stream)."""
import ctypes as C

import numpy as np

from ._lib import TRANS_DTYPE, check, lib, ptr


def _dptr(t):
    """device pointer of a torch tensor (or an int that already is one)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class Plan:
    """One recording channel of length T with a fixed model shape."""

    def __init__(self, T, lA, mu, sigma):
        self._h = C.c_void_p(None)
        self.T = int(T)
        self.lA = lA
        mu = np.asfortranarray(mu, dtype=np.float64)
        tr = np.ascontiguousarray(lA.transitions, dtype=TRANS_DTYPE)
        st = np.asfortranarray(lA.states, dtype=np.int16)
        check(lib().hmmsort_plan_create(C.byref(self._h), self.T, ptr(st), lA.N, lA.K, lA.nstates,
                                        ptr(tr), len(tr), ptr(mu), float(sigma)))
        self.K, self.N, self.S = lA.K, lA.N, lA.nstates

    @classmethod
    def batched(cls, T, lAs, mus, sigmas):
        """C channels of length T with per-channel models of one shape (hmmsort_plan_create_batched).
        Device buffers of every call are channel-major: y [C][T], x [C][T], ll [C], stats [C][len]."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(None)
        self.T = int(T)
        lA = lAs[0]
        self.lA = lA
        nC = len(lAs)
        R = len(lA.transitions)
        tr = np.ascontiguousarray(np.stack([np.ascontiguousarray(a.transitions, dtype=TRANS_DTYPE) for a in lAs]))
        assert tr.shape == (nC, R), "batched plan: every channel needs the same transition count"
        mu = np.ascontiguousarray(np.stack([np.asfortranarray(m, dtype=np.float64).ravel(order="F") for m in mus]))
        sg = np.ascontiguousarray(sigmas, dtype=np.float64)
        st = np.asfortranarray(lA.states, dtype=np.int16)
        check(lib().hmmsort_plan_create_batched(C.byref(self._h), nC, self.T, ptr(st), lA.N, lA.K, lA.nstates,
                                                ptr(tr), R, ptr(mu), ptr(sg)))
        self.K, self.N, self.S = lA.K, lA.N, lA.nstates
        self.C = nC
        return self

    def channels(self):
        return int(lib().hmmsort_plan_channels(self._h))

    def set_model_channel(self, ch, lA, mu, sigma):
        mu = np.asfortranarray(mu, dtype=np.float64)
        tr = np.ascontiguousarray(lA.transitions, dtype=TRANS_DTYPE)
        check(lib().hmmsort_plan_set_model_channel(self._h, int(ch), ptr(tr), len(tr), ptr(mu), float(sigma)))

    def close(self):
        if self._h:
            lib().hmmsort_plan_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def info(self):
        v = [C.c_int64(0) for _ in range(5)]
        check(lib().hmmsort_plan_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("engine", "block", "halo", "nchains", "workspace_bytes"),
                        [x.value for x in v]))

    def overlap_sweep(self):
        """which sweep an overlap model runs on the blocked engine: 0 generic, 2 pair sweep, 3..5 multi sweep"""
        return int(lib().hmmsort_plan_overlap_sweep(self._h))

    def set_model(self, lA, mu, sigma):
        mu = np.asfortranarray(mu, dtype=np.float64)
        tr = np.ascontiguousarray(lA.transitions, dtype=TRANS_DTYPE)
        check(lib().hmmsort_plan_set_model(self._h, ptr(tr), len(tr), ptr(mu), float(sigma)))
        self.lA = lA

    def bind(self, d_y, stream=0):
        """transpose + ring-score pre-pass once; following viterbi/estep calls on d_y reuse them"""
        check(lib().hmmsort_plan_bind(self._h, _dptr(d_y), C.c_void_p(stream)))

    def unbind(self):
        check(lib().hmmsort_plan_unbind(self._h))

    def viterbi(self, d_y, d_x, d_ll, stream=0):
        check(lib().hmmsort_plan_viterbi(self._h, _dptr(d_y), _dptr(d_x), _dptr(d_ll),
                                         C.c_void_p(stream)))

    def set_shard(self, own_lo, own_hi, first, last):
        """time shard of a longer recording: accumulate statistics for [own_lo, own_hi) only"""
        check(lib().hmmsort_plan_set_shard(self._h, int(own_lo), int(own_hi), int(bool(first)),
                                           int(bool(last))))

    def stats_len(self):
        return int(lib().hmmsort_plan_stats_len(self._h))

    def mstep_len(self):
        """doubles per channel of the M-step output [mu (K x N) | sigma | xb[2:end] | pp (S)], as the library
        writes them (hmmsort_plan_mstep_len): N entry log-probabilities for wave/ring plans whatever the
        list has dropped (types.jl:121), one per transition leaving state 1 but the first otherwise
        (baumwelch.jl:226,264)"""
        return int(lib().hmmsort_plan_mstep_len(self._h))

    def estep(self, d_y, d_stats, stream=0):
        check(lib().hmmsort_plan_estep(self._h, _dptr(d_y), _dptr(d_stats), C.c_void_p(stream)))

    def decode_estep(self, d_y, d_x, d_ll, d_stats, stream=0):
        """viterbi + estep of the same signal/model with the three sweeps sharing one launch"""
        check(lib().hmmsort_plan_decode_estep(self._h, _dptr(d_y), _dptr(d_x), _dptr(d_ll),
                                              _dptr(d_stats), C.c_void_p(stream)))

    def mstep(self, d_stats, d_out, stream=0):
        check(lib().hmmsort_plan_mstep(self._h, _dptr(d_stats), _dptr(d_out), C.c_void_p(stream)))

    def diagnostics(self, stream=0):
        d = (C.c_int64 * 8)()
        check(lib().hmmsort_plan_diagnostics(self._h, C.c_void_p(stream), d))
        out = list(d)
        import struct
        for i in (2, 4, 6):  # largest boundary errors travel as double bit patterns
            out[i] = struct.unpack("<d", struct.pack("<q", out[i]))[0]
        return out

    def tie_stats(self, stream=0):
        """what the exact near-tie resolver did in the last decode (hmmsort_plan_tie_stats)"""
        d = (C.c_int64 * 8)()
        check(lib().hmmsort_plan_tie_stats(self._h, C.c_void_p(stream), d))
        return dict(zip(("trigger", "flagged", "decided", "flips", "unresolved", "tail", "longest_walk",
                         "serial_blocks"), list(d)))

    def reconstruct(self, d_x, d_y_out, stream=0):
        """reconstruct_signal of a decoded path, device to device (T doubles)"""
        check(lib().hmmsort_plan_reconstruct(self._h, _dptr(d_x), _dptr(d_y_out), C.c_void_p(stream)))

    def unroll_mlseq(self, d_x, d_out, stream=0):
        """unroll_mlseq of a decoded path, device to device (N x T int16, column-major)"""
        check(lib().hmmsort_plan_unroll_mlseq(self._h, _dptr(d_x), _dptr(d_out), C.c_void_p(stream)))

    def extract_spiketimes(self, d_x, stream=0):
        """extract_spiketimes (extraction.jl:15-24) from the decoded path in device memory:
        list of 1-based sample-index arrays, one per neuron (only the spike times leave the GPU)."""
        counts = np.zeros(self.N, dtype=np.int64)
        check(lib().hmmsort_plan_extract_spiketimes(self._h, _dptr(d_x), None, 0, ptr(counts),
                                                    C.c_void_p(stream)))
        cap = int(counts.max()) if self.N else 0
        times = np.zeros((self.N, max(cap, 1)), dtype=np.int64)
        if cap:
            check(lib().hmmsort_plan_extract_spiketimes(self._h, _dptr(d_x), ptr(times), cap,
                                                        ptr(counts), C.c_void_p(stream)))
        return [times[i, :counts[i]].copy() for i in range(self.N)]

    def profile(self, enable=True):
        check(lib().hmmsort_plan_profile(self._h, int(bool(enable))))

    def profile_read(self, stream=0):
        """{kernel name: (total ms, launches)} since the previous read (synchronises the stream)."""
        cap = 64
        names = C.create_string_buffer(4096)
        ms = (C.c_double * cap)()
        calls = (C.c_int64 * cap)()
        n = C.c_int64(0)
        check(lib().hmmsort_plan_profile_read(self._h, C.c_void_p(stream), names, 4096, ms, calls,
                                              cap, C.byref(n)))
        nm = names.value.decode().split("\n") if n.value else []
        return {nm[i]: (ms[i], calls[i]) for i in range(n.value)}
