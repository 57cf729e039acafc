"""ctypes binding of libhmmsort_hip.so (the C ABI declared in include/hmmsort.h).

There is no fallback of any kind: if the shared library is missing this module raises, and if no
HIP device is present every compute entry point returns HMMSORT_EHIP, surfaced as HmmsortError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMMSORT_LIB", os.path.join(_HERE, "libhmmsort_hip.so"))  # override: A/B builds

# Julia Tuple{Int64,Int64,Float64} == struct hmm_trans (include/hmmsort.h)
TRANS_DTYPE = np.dtype([("src", np.int64), ("dst", np.int64), ("lp", np.float64)], align=True)
assert TRANS_DTYPE.itemsize == 24

OK, EINVAL, ENOMEM, EHIP, ENOCONV, EUNSUP = 0, -1, -2, -3, -4, -5
ENGINE_AUTO, ENGINE_STRICT, ENGINE_RING, ENGINE_BLOCKED, ENGINE_WAVE = 0, 1, 2, 3, 4


class HmmsortError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hmmsort error %d: %s" % (code, msg))
        self.code = code


_i64, _f64, _int = C.c_int64, C.c_double, C.c_int
_vp = C.c_void_p
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/hmmsort.h declares
SIGNATURES = {
    "hmmsort_last_error": (C.c_char_p, []),
    "hmmsort_version": (_int, []),
    "hmmsort_device_count": (_int, [C.POINTER(_int)]),
    "hmmsort_set_device": (_int, [_int]),
    "hmmsort_set_option": (_int, [C.c_char_p, _i64]),
    "hmmsort_get_option": (_int, [C.c_char_p, _pi64]),
    "hmmsort_shutdown": (_int, []),
    "hmmsort_generate_states": (_i64, [_i64, _i64, _int, _vp]),
    "hmmsort_build_transitions": (_i64, [_i64, _i64, _vp, _i64, _int, _vp, _i64]),
    "hmmsort_viterbi": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _vp, _vp]),
    "hmmsort_viterbi_i16": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _vp, _vp]),
    "hmmsort_samples_to_f64": (_int, [_vp, C.c_int, _i64, _i64, _vp, _vp]),
    "hmmsort_forward": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _vp]),
    "hmmsort_backward": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _vp]),
    "hmmsort_update": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64,
                              _vp, _vp, _i64, _pi64, _vp]),
    "hmmsort_em_step": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _f64, _vp, _vp,
                               _i64, _pi64, _vp]),
    "hmmsort_reconstruct": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "hmmsort_unroll_mlseq": (_int, [_vp, _i64, _vp, _i64, _i64, _vp]),
    "hmmsort_extract_spiketimes": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "hmmsort_plan_create": (_int, [C.POINTER(_vp), _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp,
                                   _f64]),
    "hmmsort_plan_set_model": (_int, [_vp, _vp, _i64, _vp, _f64]),
    "hmmsort_plan_create_batched": (_int, [C.POINTER(_vp), _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64,
                                           _vp, _vp]),
    "hmmsort_plan_channels": (_i64, [_vp]),
    "hmmsort_plan_set_model_channel": (_int, [_vp, _i64, _vp, _i64, _vp, _f64]),
    "hmmsort_plan_destroy": (_int, [_vp]),
    "hmmsort_plan_info": (_int, [_vp, _pi64, _pi64, _pi64, _pi64, _pi64]),
    "hmmsort_plan_overlap_sweep": (_i64, [_vp]),
    "hmmsort_plan_bind": (_int, [_vp, _vp, _vp]),
    "hmmsort_plan_unbind": (_int, [_vp]),
    "hmmsort_plan_viterbi": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "hmmsort_plan_estep": (_int, [_vp, _vp, _vp, _vp]),
    "hmmsort_plan_decode_estep": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "hmmsort_plan_stats_len": (_i64, [_vp]),
    "hmmsort_plan_set_shard": (_int, [_vp, _i64, _i64, _int, _int]),
    "hmmsort_plan_mstep": (_int, [_vp, _vp, _vp, _vp]),
    "hmmsort_plan_mstep_len": (_i64, [_vp]),
    "hmmsort_plan_diagnostics": (_int, [_vp, _vp, _pi64]),
    "hmmsort_plan_tie_stats": (_int, [_vp, _vp, _pi64]),
    "hmmsort_plan_extract_spiketimes": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "hmmsort_plan_reconstruct": (_int, [_vp, _vp, _vp, _vp]),
    "hmmsort_plan_unroll_mlseq": (_int, [_vp, _vp, _vp, _vp]),
    "hmmsort_plan_profile": (_int, [_vp, _int]),
    "hmmsort_plan_profile_read": (_int, [_vp, _vp, C.c_char_p, _i64, C.POINTER(_f64), _pi64, _i64,
                                         _pi64]),
}

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libhmmsort_hip.so is not built (%s). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s/csrc`; "
                "there is no CPU fallback." % (LIB_PATH, _HERE))
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 and fails to
        # find the GPU if the system copy (same soname) was mapped first.  Importing torch first
        # makes this library bind to the runtime torch uses; without torch the system ROCm copy
        # (RUNPATH /opt/rocm/lib) is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error():
    return lib().hmmsort_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise HmmsortError(rc, last_error())


def ptr(a):
    """void* of a numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    check(lib().hmmsort_device_count(C.byref(n)))
    return n.value


def set_option(key, value):
    check(lib().hmmsort_set_option(key.encode(), int(value)))


def shutdown():
    """hmmsort_shutdown: free the idle plans and device buffers the host-buffer entry points keep"""
    check(lib().hmmsort_shutdown())


def get_option(key):
    v = C.c_int64(0)
    check(lib().hmmsort_get_option(key.encode(), C.byref(v)))
    return v.value
