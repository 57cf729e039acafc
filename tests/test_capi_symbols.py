"""The C-ABI library loads and exports every symbol include/hmmsort.h declares; the ctypes
signature table covers exactly those symbols.  No compute calls (CPU only)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hmmsort.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmmsort_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(H):
    names = _declared()
    assert len(names) >= 20
    L = ctypes.CDLL(H._lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export: " + n


def test_signature_table_matches_header(H):
    assert sorted(H._lib.SIGNATURES) == _declared()


def test_trans_record_layout(H):
    # Julia Tuple{Int64,Int64,Float64} is 24 bytes, fields at 0/8/16
    dt = H._lib.TRANS_DTYPE
    assert dt.itemsize == 24 and [dt.fields[k][1] for k in ("src", "dst", "lp")] == [0, 8, 16]


def test_no_cpu_fallback(H):
    import numpy as np
    import pytest
    if H.device_count() > 0:
        pytest.skip("GPU present")
    sm = H.StateMatrix.create(2, 5, np.log([0.01, 0.004]), False)
    with pytest.raises(H.HmmsortError) as e:
        H.viterbi(np.zeros(50), sm, np.zeros((5, 2)), 0.3)
    assert e.value.code == H._lib.EHIP


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hmmspikesorter.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower().replace("the cpu oracle", ""), f
