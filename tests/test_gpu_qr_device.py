"""cs_happly for blocks of vectors on the device (csx_happly) and the device solve phase of cs_qrsol built on it,
against the oracle's cs_happly / the drop-in's own list-level cs_qrsol, bit for bit (needs an MI355X)."""
import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack
from test_gpu_parity import _host_cs, cs  # noqa: F401  (cs is the module fixture)

pytestmark = pytest.mark.gpu


def _factor(cs, name):
    g = golden(name)
    A = unpack(cs, g, "A")
    S = cs.cs_sqr(0, A, True)
    N = cs.cs_qr(A, S)
    assert N is not None
    return g, A, S, N


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "ash219", "lp_afiro"])
@pytest.mark.parametrize("transpose", [True, False])
def test_happly_block_bit_identical(cs, name, transpose):
    g, A, S, N = _factor(cs, name) if golden(name)["A_mn"][0] >= golden(name)["A_mn"][1] else (None, None, None, None)
    if N is None:
        pytest.skip("m < n: cs_qrsol factors the transpose")
    V, n, m2, k = N.L, N.L.n, N.L.m, 70                         # more vectors than one wave
    rng = np.random.default_rng(7)
    B = rng.uniform(-1, 1, size=(m2, k))
    X = cs.dvec(B)
    assert cs.apply_q(N, X, transpose) is True
    got = X.numpy()
    oV = O.cs_spalloc(m2, n, len(V.i), True, False)
    oV.p, oV.i, oV.x = list(V.p), list(V.i), list(V.x)
    for r in (0, 1, 63, 64, k - 1):
        x = B[:, r].tolist()
        for t in range(n):
            i = t if transpose else n - 1 - t
            O.cs_happly(oV, i, N.B[i], x)
        assert got[:, r].tobytes() == np.asarray(x).tobytes(), (name, r)


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "ash219"])
def test_qrsol_solver_matches_list_level_qrsol(cs, name):
    g = golden(name)
    A = unpack(cs, g, "A")
    if A.m < A.n:
        pytest.skip("m < n")
    F = cs.qrsol_factor(A)
    assert F is not None
    m, n, k = A.m, A.n, 5
    rng = np.random.default_rng(3)
    B = rng.uniform(-1, 1, size=(m, k))
    X = F.solve(cs.dvec(B)).numpy().reshape(n, k)
    for r in range(k):
        b = B[:, r].tolist() + [0.0] * max(0, n - m)
        assert cs.cs_qrsol(0, A, b) is True
        assert X[:, r].tobytes() == np.asarray(b[:n]).tobytes(), (name, r)
    one = B[:, 0].tolist()                                      # a list right-hand side: overwritten like cs_qrsol's b
    assert F.solve(one) is True
    ref = B[:, 0].tolist()
    assert cs.cs_qrsol(0, A, ref) is True and one[:n] == ref[:n]
