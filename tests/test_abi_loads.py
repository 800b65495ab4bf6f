"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/csx.h declares, and the host-only helpers of the product
module behave like the reference.  No GPU work here."""
import os
import re

import pytest

from conftest import ROOT, golden, same_csc


def header_symbols():
    text = open(os.path.join(ROOT, "include", "csx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import _csx
    lib = _csx.load()
    names = header_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_csx.exported_symbols()) == names


def test_product_does_not_touch_oracle():
    pkg = os.path.join(ROOT, "csparse.py_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), f


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import _csx
    import csparse as cs
    A = cs.cs_spalloc(2, 2, 2, True, False)
    A.p, A.i, A.x = [0, 1, 2], [0, 1], [1.0, 2.0]
    with pytest.raises(_csx.CsxError):
        cs.cs_gaxpy(A, [1.0, 1.0], [0.0, 0.0])


def test_host_glue_matches_reference_vectors():
    import csparse as cs
    g = golden("west0067")
    m, n, nzmax, nz = (int(v) for v in g["T_mn"])
    T = cs.cs_spalloc(0, 0, 1, True, True)
    for i, j, x in zip(g["T_i"], g["T_j"], g["T_x"]):
        assert cs.cs_entry(T, int(i), int(j), float(x))
    assert (T.m, T.n, T.nzmax, T.nz) == (m, n, nzmax, nz)
    A = cs.cs_compress(T)
    same_csc(A, g, "A")
    assert cs.cs_norm(A) == 6.1433746000000005
    s = golden("synthetic_20240601")
    c = [int(v) for v in s["cumsum_c"]]
    p = [0] * (len(c) + 1)
    assert cs.cs_cumsum(p, c, len(c)) == int(s["cumsum_ret"][0])
    assert p == s["cumsum_p"].tolist() and c == s["cumsum_c_out"].tolist()
    perm, b = s["perm"].tolist(), s["perm_b"].tolist()
    x = [0.0] * len(b)
    assert cs.cs_ipvec(perm, b, x, len(b)) and x == s["ipvec"].tolist()
    assert cs.cs_pvec(perm, b, x, len(b)) and x == s["pvec"].tolist()
    assert cs.cs_ipvec(None, b, x, len(b)) and x == b
    assert cs.cs_pvec(perm, None, x, 3) is False and cs.cs_cumsum(None, c, 1) == -1
    # cs_scatter on lists: rebuild one column of A*A' the way cs_multiply does
    AT = golden("west0067")
    from conftest import unpack
    At = unpack(cs, AT, "AT")
    C = cs.cs_spalloc(A.m, At.n, A.m, True, False)
    w, xx = [0] * A.m, [0.0] * A.m
    nzc = 0
    for pp in range(At.p[0], At.p[1]):
        nzc = cs.cs_scatter(A, At.i[pp], At.x[pp], w, xx, 1, C, nzc)
    assert C.i[:nzc] == AT["AAT_i"][:nzc].tolist()
    assert [xx[r] for r in C.i[:nzc]] == AT["AAT_x"][:nzc].tolist()
    assert cs.cs_scatter(T, 0, 1.0, w, xx, 1, C, 0) == -1


def test_error_conventions_without_gpu():
    import csparse as cs
    T = cs.cs_spalloc(3, 3, 1, True, True)
    assert cs.cs_gaxpy(T, [0.0] * 3, [0.0] * 3) is False
    assert cs.cs_gaxpy(None, [], []) is False
    assert cs.cs_transpose(T, True) is None and cs.cs_multiply(T, T) is None
    assert cs.cs_lsolve(T, [0.0]) is False and cs.cs_usolve(None, [0.0]) is False
    assert cs.cs_ltsolve(T, None) is False and cs.cs_utsolve(None, None) is False
    A = cs.cs_spalloc(2, 3, 1, True, False)
    B = cs.cs_spalloc(2, 3, 1, True, False)
    assert cs.cs_multiply(A, B) is None  # inner dimensions differ (csparse.py:1618)
    assert cs.cs_cholsol(0, T, [0.0]) is False and cs.cs_cholsol(0, A, None) is False
    assert cs.cs_chol(A, None) is None and cs.cs_schol(0, T) is None
