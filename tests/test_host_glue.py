"""The reference's inner helpers provided as host code: cs_dfs / cs_reach / cs_spsolve /
cs_ereach of the product against the oracle (which is pinned to the unmodified reference), on the golden
matrices and on random triangles; marks must be undone on return."""
import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack

import _hostglue as H


def _tri(rng, n, mean, lower):
    A = O.cs_spalloc(n, n, 1, True, True)
    for j in range(n):
        O.cs_entry(A, j, j, float(rng.uniform(2, 3)))
        lo, hi = (j + 1, n) if lower else (0, j)
        k = min(int(rng.poisson(mean)), hi - lo)
        for i in (rng.choice(np.arange(lo, hi), size=k, replace=False) if k else []):
            O.cs_entry(A, int(i), j, float(rng.uniform(-1, 1)))
    C = O.cs_compress(A)
    if lower:           # diagonal first
        return C
    T = O.cs_transpose(O.cs_transpose(C, True), True)   # sorted columns: diagonal last
    return T


def _sparse_rhs(rng, n, k):
    B = O.cs_spalloc(n, 1, 1, True, True)
    for i in rng.choice(n, size=k, replace=False):
        O.cs_entry(B, int(i), 0, float(rng.uniform(-1, 1)))
    return O.cs_compress(B)


@pytest.mark.parametrize("lower", [True, False])
@pytest.mark.parametrize("n,mean,k", [(1, 0, 1), (30, 2.0, 3), (200, 1.5, 5), (200, 0.3, 40)])
def test_reach_and_spsolve_match_oracle(n, mean, k, lower):
    rng = np.random.default_rng(n * 17 + k + int(lower))
    G = _tri(rng, n, mean, lower)
    B = _sparse_rhs(rng, n, min(k, n))
    p_before = list(G.p)
    xi1, xi2 = [0] * (2 * n), [0] * (2 * n)
    x1, x2 = [0.0] * n, [0.0] * n
    t1 = H.cs_spsolve(G, B, 0, xi1, x1, None, lower)
    assert list(G.p) == p_before                      # marks undone
    t2 = O.cs_spsolve(G, B, 0, xi2, x2, None, lower)
    assert t1 == t2 and xi1[t1:n] == xi2[t2:n]
    assert [x1[i] for i in xi1[t1:n]] == [x2[i] for i in xi2[t2:n]]
    assert H.cs_reach(G, B, 0, [0] * (2 * n), None) == t2 and list(G.p) == p_before
    # a row permutation with not-yet-pivotal rows, as cs_lu uses it
    pinv = rng.permutation(n).tolist()
    for i in rng.choice(n, size=n // 3, replace=False):
        pinv[int(i)] = -1
    a, b = [0] * (2 * n), [0] * (2 * n)
    assert H.cs_reach(G, B, 0, a, pinv) == O.cs_reach(G, B, 0, b, pinv)
    assert list(G.p) == p_before


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_ereach_matches_golden_and_oracle(name):
    g = golden(name)
    C = unpack(O, g, "C")
    Cu = O.cs_symperm(C, None, False)
    parent = O.cs_etree(Cu, False)
    n = C.n
    w1, s1, w2, s2 = [0] * n, [0] * n, [0] * n, [0] * n
    for k in range(0, n, max(1, n // 97)):
        t1 = H.cs_ereach(Cu, k, parent, s1, 0, w1)
        t2 = O.cs_ereach(Cu, k, parent, s2, 0, w2)
        assert t1 == t2 and s1[t1:] == s2[t2:] and w1 == [0] * n
    if "ereach_top" in g:                              # rows recorded from the unmodified reference
        tops = [int(v) for v in g["ereach_top"]]
        rows = [int(v) for v in g["ereach_k"]] if "ereach_k" in g else None
        if rows is not None:
            for k, t in zip(rows, tops):
                assert H.cs_ereach(Cu, k, parent, s1, 0, w1) == t


def test_bad_arguments_and_macros():
    assert H.CS_FLIP(3) == -5 and H.CS_UNFLIP(-5) == 3 and H.CS_UNFLIP(4) == 4
    w = [0, 7]
    H.CS_MARK(w, 1)
    assert H.CS_MARKED(w, 1) and not H.CS_MARKED(w, 0)
    assert H.cs_spsolve(None, None, 0, [0], [0.0], None, True) == -1
    assert H.cs_ereach(None, 0, [0], [0], 0, [0]) == -1 and H.cs_reach(None, None, 0, [0], None) == -1


def test_names_are_in_the_drop_in_module():
    import csparse as cs
    for name in ("cs_dfs", "cs_reach", "cs_spsolve", "cs_ereach", "CS_FLIP", "CS_UNFLIP", "CS_MARKED", "CS_MARK"):
        assert getattr(cs, name) is getattr(H, name)
