"""Orders 1 - 3 of cs_lusol and order 3 of cs_qrsol (csparse.py:1456-1478, :1875-1912, :2187-2217; the calls of
csparse_test.py:456-482).  The reference's cs_amd does not run, so there is no permutation to match: the
ordering here is a nested dissection of the same graph (A + A', S'S, A'A).  What the reference's tests pin is the
solution: ||x||_inf per matrix (csparse_test.py:496-642, absolute delta 1e-3), the same for every order because the
solution is unique.  Checked here together with the residual of the permuted factorisation, and against the order-0
answers of the unmodified reference where those exist (tests/golden: x_lusol, x_qrsol)."""
import numpy as np
import pytest

import tol as TOL

import c_oracle as CO
from conftest import golden, unpack
from test_gpu_parity import cs  # noqa: F401

pytestmark = pytest.mark.gpu

# csparse_test.py Test2 known answers
NORM = {"t1": 2.4550, "bcsstk01": 0.0005, "west0067": 21.9478, "fs_183_1": 212022.2099, "bcsstk16": 1.9998,
        "ash219": 1.0052, "ibm32a": 5.5800, "ibm32b": 5.3348, "lp_afiro": 2.4534}


def _rhs(m):
    return [1.0 + float(i) / m for i in range(m)]        # csparse_test.py:123-127


def _arrays(C):
    nnz = C.p[C.n]
    return np.asarray(C.p, np.int32), np.asarray(C.i[:nnz], np.int32), np.asarray(C.x[:nnz])


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1", "bcsstk16"])
def test_lusol_with_a_fill_reducing_ordering(cs, name, order, meta):
    g = golden(name)
    C = unpack(cs, g, "C")
    n = C.n
    tol = 0.001 if meta[name]["sym"] else 1.0              # csparse_test.py:443
    q = cs.cs_amd(order, C)
    assert sorted(q) == list(range(n))                     # a permutation of the columns
    S = cs.cs_sqr(order, C, False)
    assert S is not None and sorted(S.q) == list(range(n)) and S.lnz == 4 * C.p[n] + n
    b = _rhs(n)
    alias = b
    assert cs.cs_lusol(order, C, b, tol) is True and alias is b
    x = np.asarray(b)
    assert np.max(np.abs(x)) == pytest.approx(NORM[name], abs=1e-3)
    Cp, Ci, Cx = _arrays(C)
    r = CO.gaxpy(n, n, Cp, Ci, Cx, x, -np.asarray(_rhs(n)))
    scale = np.max(np.abs(Cx)) * np.max(np.abs(x)) * 64 + 2.0
    assert np.max(np.abs(r)) <= 1e-12 * scale
    if "x_lusol" in g:                                     # the unmodified reference's order-0 answer: the same solution
        ref = g["x_lusol"]                                 # another pivot sequence = another factorisation: the matrix's conditioning
        assert TOL.normwise(x, ref) <= TOL.cross_bound(TOL.cond1(TOL.csc(n, Cp, Ci, Cx)))


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("name", ["west0067", "fs_183_1"])
def test_lusol_factor_solves_lists_and_blocks_with_the_bits_of_cs_lusol(cs, name, order, meta):
    """lusol_factor (factor once; permute, L, U, permute on the device) against the driver cs_lusol, column by column, at
    every ordering: the same factors, the same operation order, so the same bits."""
    g = golden(name)
    C = cs.cs_pin(unpack(cs, g, "C"))
    n, k = C.n, 5
    tol = 0.001 if meta[name]["sym"] else 1.0
    F = cs.lusol_factor(C, order, tol)
    B = np.stack([np.asarray(_rhs(n)) * (1.0 + 0.25 * r) + r for r in range(k)], axis=1)
    dB = cs.dvec(np.ascontiguousarray(B))
    assert F.solve(dB) is True
    X = dB.numpy().reshape(n, k)
    for r in range(k):
        col = B[:, r].tolist()
        assert cs.cs_lusol(order, C, col, tol) is True
        assert np.asarray(col).tobytes() == np.ascontiguousarray(X[:, r]).tobytes(), (name, order, r)
    one = B[:, 2].tolist()
    assert F.solve(one) is True and np.asarray(one).tobytes() == np.ascontiguousarray(X[:, 2]).tobytes()
    blk = cs.dvec(np.ascontiguousarray(B))
    assert cs.cs_lusol(order, C, blk, tol) is True and blk.numpy().tobytes() == dB.numpy().tobytes()
    assert cs.lusol_factor(unpack(cs, golden("ash219"), "C")) is None          # not square


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1", "ash219", "ibm32a", "ibm32b", "lp_afiro"])
def test_qrsol_order_3(cs, name):
    g = golden(name)
    C = unpack(cs, g, "C")
    m, n = C.m, C.n
    k = min(m, n)
    S = cs.cs_sqr(3, C if m >= n else cs.cs_transpose(C, True), True)
    assert S is not None and sorted(S.q) == list(range(k)) and S.m2 >= max(m, n) - 0
    b = _rhs(m) + [0.0] * max(0, n - m)
    assert cs.cs_qrsol(3, C, b) is True
    x = np.asarray(b[:n])
    assert np.max(np.abs(x)) == pytest.approx(NORM[name], abs=1e-3)
    b0 = _rhs(m) + [0.0] * max(0, n - m)
    assert cs.cs_qrsol(0, C, b0) is True                   # the same least-squares / minimum-norm solution as order 0
    Cp, Ci, Cx = _arrays(C)
    # two orderings = two QR factorisations: a least-squares / minimum-norm solution moves with cond_2(A)^2 (tests/tol.py)
    assert TOL.normwise(x, np.asarray(b0[:n])) <= TOL.lsq_bound(TOL.cond2_dense(TOL.csc(n, Cp, Ci, Cx, m)))
    r = CO.gaxpy(m, n, Cp, Ci, Cx, x, -np.asarray(_rhs(m)))
    if m >= n:                                             # normal equations: A'(A x - b) = 0
        Tp, Ti, Tx = CO.transpose(m, n, Cp, Ci, Cx)
        g0 = CO.gaxpy(n, m, Tp, Ti, Tx, r, np.zeros(n))
        assert np.max(np.abs(g0)) <= 1e-9 * np.max(np.abs(Cx)) ** 2 * max(np.max(np.abs(x)), 1.0) * m
    else:
        assert np.max(np.abs(r)) < 1e-10


def test_amd_conventions(cs):
    g = golden("west0067")
    C = unpack(cs, g, "C")
    assert cs.cs_amd(0, C) is None and cs.cs_amd(4, C) is None and cs.cs_amd(-1, C) is None and cs.cs_amd(1, None) is None
    T = cs.cs_spalloc(2, 2, 2, True, True)
    assert cs.cs_amd(1, T) is None and cs.cs_sqr(1, T, False) is None
    R = unpack(cs, golden("ash219"), "C")                  # 219 x 85: order 1 falls back to the graph of A'A, like orders 2 / 3
    for order in (1, 2, 3):
        q = cs.cs_amd(order, R)
        assert sorted(q) == list(range(R.n))
