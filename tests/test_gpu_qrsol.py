"""cs_qrsol (csparse.py:1875-1912): host Householder QR + device cs_usolve / cs_utsolve.
Square systems are pinned by the unmodified reference's own cs_qrsol(0, ...) output; for
rectangular matrices the reference's port is off and the pin is the expected value
in the reference's test file (csparse_test.py:496, :564, :578, :592; absolute delta 1e-3)."""
import numpy as np
import pytest

import tol as TOL
from conftest import golden, unpack
from test_gpu_parity import RTOL, cs  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_qrsol_square_matches_reference(cs, name, meta):
    g = golden(name)
    C = unpack(cs, g, "C")
    b = g["b"].tolist()
    alias = b
    assert cs.cs_qrsol(0, C, b) is True and alias is b
    ref = g["x_qrsol"]
    assert TOL.normwise(b, ref) < TOL.X_RTOL       # same algorithm, same operation order as the reference's cs_qrsol
    assert max(abs(v) for v in b) == pytest.approx(meta[name]["qrsol_norm_inf"], rel=TOL.X_RTOL)
    # and it agrees with the LU answer of the same system: another factorisation, so to the conditioning of the matrix
    A = TOL.csc(C.n, g["C_p"], g["C_i"], g["C_x"])
    assert TOL.normwise(b, g["x_lusol"]) < TOL.cross_bound(TOL.cond1(A))


EXPECTED = {"ash219": 1.0052, "ibm32a": 5.5800, "ibm32b": 5.3348, "lp_afiro": 2.4534}


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_qrsol_rectangular_known_answers(cs, name):
    g = golden(name)
    C = unpack(cs, g, "C")
    m, n = C.m, C.n
    b = [1.0 + float(i) / m for i in range(m)] + [0.0] * max(0, n - m)   # csparse_test.py:123, :441
    assert cs.cs_qrsol(0, C, b) is True
    x = np.asarray(b[:n])
    assert np.max(np.abs(x)) == pytest.approx(EXPECTED[name], abs=1e-3)
    # optimality: least squares -> A'(Ax - b) = 0; minimum norm -> A x = b
    import c_oracle as CO
    Ap, Ai, Ax = np.asarray(C.p, np.int32), np.asarray(C.i[:C.p[n]], np.int32), np.asarray(C.x[:C.p[n]])
    rhs = np.asarray([1.0 + float(i) / m for i in range(m)])
    r = CO.gaxpy(m, n, Ap, Ai, Ax, x, -rhs)
    if m >= n:
        Tp, Ti, Tx = CO.transpose(m, n, Ap, Ai, Ax)
        assert np.max(np.abs(CO.gaxpy(n, m, Tp, Ti, Tx, r, np.zeros(n)))) < 1e-10
    else:
        assert np.max(np.abs(r)) < 1e-10


def test_qrsol_error_conventions(cs):
    T = cs.cs_spalloc(2, 2, 2, True, True)
    assert cs.cs_qrsol(0, T, [1.0, 1.0]) is False
    A = cs.cs_spalloc(2, 2, 2, True, False)
    A.p, A.i, A.x = [0, 1, 2], [0, 1], [1.0, 2.0]
    assert cs.cs_qrsol(0, A, None) is False and cs.cs_qrsol(7, A, [1.0, 1.0]) is False
    b = [2.0, 2.0]
    assert cs.cs_qrsol(0, A, b) is True and b == [2.0, 1.0]


def test_qr_symbolic_against_reference_identities(cs):
    """cs_sqr(qr=True) pieces on the host: R's column counts bound nnz(R) of the numeric QR,
    pinv is a permutation of m2 rows, and V/R have the CSparse shapes (diagonal last in R)."""
    for name in ("west0067", "ash219", "ibm32a"):
        g = golden(name)
        C = unpack(cs, g, "C")
        S = cs.cs_sqr(0, C, True)
        assert sorted(S.pinv[:S.m2]) == list(range(S.m2)) and S.m2 >= C.m
        N = cs.cs_qr(C, S)
        n = C.n
        assert N.U.p[n] <= S.unz and N.L.p[n] <= S.lnz
        assert all(N.U.i[N.U.p[j + 1] - 1] == j for j in range(n))      # R(j,j) last in column j
        assert all(N.L.i[N.L.p[j]] == j for j in range(n))              # V(j,j) first
        # |R(j,j)| are the column norms seen by the Householder steps: R'R = A'A on the diagonal sum
        colsq = [sum(v * v for v in C.x[C.p[j]:C.p[j + 1]]) for j in range(n)]
        rsq = [sum(v * v for v in N.U.x[N.U.p[j]:N.U.p[j + 1]]) for j in range(n)]
        np.testing.assert_allclose(rsq, colsq, rtol=1e-10)
