"""Seeded random shapes against the plain-C oracle: every SpMV plan, transpose, product, the four
triangular solves (single and blocked right-hand sides).  Matrices are ragged on purpose: empty and
very long columns, duplicate rows inside a column, explicit zeros, rectangular shapes, sizes that are
not multiples of any tile."""
import numpy as np
import pytest

import c_oracle as CO
from test_gpu_parity import RTOL, _host_cs, cs, rel_err  # noqa: F401

pytestmark = pytest.mark.gpu


def ragged(rng, m, n, mean_len, dup=True):
    lens = rng.poisson(mean_len, size=n)
    lens[rng.random(n) < 0.15] = 0
    if n:
        lens[rng.integers(0, n)] = min(m * 2, int(mean_len * 20) + 3)       # one very long column
    if not dup:
        lens = np.minimum(lens, m)
    p = np.zeros(n + 1, np.int32)
    p[1:] = np.cumsum(lens)
    nnz = int(p[-1])
    if dup:
        i = rng.integers(0, m, size=nnz).astype(np.int32)
    else:
        i = np.concatenate([rng.choice(m, size=int(k), replace=False) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    x = rng.uniform(-2, 2, size=nnz)
    x[rng.random(nnz) < 0.05] = 0.0
    return p, i, x


CASES = [(1, 1, 1.0), (7, 3, 2.0), (3, 9, 1.0), (257, 255, 6.0), (1000, 1, 30.0), (1, 300, 0.7), (4099, 4097, 9.0),
         (9001, 700, 40.0), (700, 9001, 3.0), (20011, 20011, 5.0)]


@pytest.mark.parametrize("m,n,mean_len", CASES)
def test_gaxpy_transpose_multiply(cs, m, n, mean_len):
    rng = np.random.default_rng(m * 1000003 + n)
    Ap, Ai, Ax = ragged(rng, m, n, mean_len)
    A = _host_cs(cs, m, n, Ap, Ai, Ax)
    x = rng.uniform(-1, 1, size=n)
    y0 = rng.uniform(-1, 1, size=m)
    ref = CO.gaxpy(m, n, Ap, Ai, Ax, x, y0)
    scale = CO.gaxpy(m, n, Ap, Ai, np.abs(Ax), np.abs(x), np.abs(y0))
    y = y0.tolist()
    assert cs.cs_gaxpy(A, x.tolist(), y) is True
    assert np.asarray(y).tobytes() == ref.tobytes()                       # list call: reference order, bit for bit
    cs.cs_pin(A)
    for mode in (cs.GAXPY_WAVE, cs.GAXPY_TILED, cs.GAXPY_ATOMIC, cs.GAXPY_AUTO):
        dy = cs.dvec(y0)
        assert cs.cs_gaxpy(A, cs.dvec(x), dy, mode) is True
        assert rel_err(dy.numpy(), ref, scale) < RTOL, mode
    Tp, Ti, Tx = CO.transpose(m, n, Ap, Ai, Ax)
    AT = cs.cs_transpose(A, True)
    assert AT.p == Tp.tolist() and AT.i[:int(Tp[-1])] == Ti.tolist()
    assert np.asarray(AT.x[:int(Tp[-1])]).tobytes() == Tx.tobytes()
    if Ap[-1] and m * n <= 7_000_000 * 10:
        Cp, Ci, Cx = CO.multiply(m, n, m, Ap, Ai, Ax, Tp, Ti, Tx)
        _, _, Sx = CO.multiply(m, n, m, Ap, Ai, np.abs(Ax), Tp, Ti, np.abs(Tx))
        C = cs.cs_multiply(A, AT)
        nnz = int(Cp[-1])
        assert C.p == Cp.tolist() and C.i[:nnz] == Ci.tolist() and C.nzmax == nnz
        assert rel_err(C.x[:nnz], Cx, Sx) < RTOL


def triangular(rng, n, mean_len, lower):
    """Well-conditioned triangle in the layout the solves expect: diagonal first (L) or last (U)."""
    cols_i, cols_x = [], []
    for j in range(n):
        lo, hi = (j + 1, n) if lower else (0, j)
        k = min(int(rng.poisson(mean_len)), hi - lo)
        off = rng.choice(np.arange(lo, hi), size=k, replace=False) if k else np.zeros(0, np.int64)
        if rng.random() < 0.5:
            off = np.sort(off)
        vals = rng.uniform(-1, 1, size=k)
        d = float(rng.uniform(2.0, 4.0) * (1 + k))
        if lower:
            cols_i.append(np.concatenate([[j], off]))
            cols_x.append(np.concatenate([[d], vals]))
        else:
            cols_i.append(np.concatenate([off, [j]]))
            cols_x.append(np.concatenate([vals, [d]]))
    p = np.zeros(n + 1, np.int32)
    p[1:] = np.cumsum([len(c) for c in cols_i])
    return p, np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)


@pytest.mark.parametrize("n,mean_len", [(1, 0.0), (2, 1.0), (65, 3.0), (1000, 0.0), (3001, 2.5), (600, 40.0), (12007, 4.0),
                                        (700, 230.0)])   # columns of 65..128 and more than 128 entries: every round of k_tri_chain
def test_triangular_solves_bit_identical(cs, n, mean_len):
    rng = np.random.default_rng(n * 31 + 7)
    Lp, Li, Lx = triangular(rng, n, mean_len, True)
    Up, Ui, Ux = triangular(rng, n, mean_len, False)
    L, U = cs.cs_pin(_host_cs(cs, n, n, Lp, Li, Lx)), cs.cs_pin(_host_cs(cs, n, n, Up, Ui, Ux))
    k = 5
    B = rng.uniform(-1, 1, size=(n, k))
    for fn, ofn, M, arr in ((cs.cs_lsolve, CO.lsolve, L, (Lp, Li, Lx)), (cs.cs_ltsolve, CO.ltsolve, L, (Lp, Li, Lx)),
                            (cs.cs_usolve, CO.usolve, U, (Up, Ui, Ux)), (cs.cs_utsolve, CO.utsolve, U, (Up, Ui, Ux))):
        b = B[:, 0].tolist()
        assert fn(M, b) is True
        assert np.asarray(b).tobytes() == ofn(n, *arr, B[:, 0]).tobytes(), fn.__name__
        dB = cs.dvec(B)
        assert fn(M, dB) is True
        Xk = dB.numpy()
        for r in range(k):
            assert Xk[:, r].tobytes() == ofn(n, *arr, B[:, r]).tobytes(), (fn.__name__, r)


def test_banded_triangles_many_right_hand_sides(cs):
    """A banded (chain-like) triangle with more right-hand sides than one 64-lane chunk: the blocked chain
    walker loops over chunks of right-hand sides and must stay bit-identical in each."""
    rng = np.random.default_rng(99)
    n, k = 900, 130
    for lower in (True, False):
        cols_i, cols_x = [], []
        for j in range(n):
            lo, hi = (j + 1, min(n, j + 40)) if lower else (max(0, j - 40), j)
            off = np.arange(lo, hi)
            vals = rng.uniform(-1, 1, size=len(off))
            d = float(rng.uniform(40.0, 60.0))
            cols_i.append(np.concatenate([[j], off]) if lower else np.concatenate([off, [j]]))
            cols_x.append(np.concatenate([[d], vals]) if lower else np.concatenate([vals, [d]]))
        p = np.zeros(n + 1, np.int32)
        p[1:] = np.cumsum([len(c) for c in cols_i])
        i, x = np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)
        T = cs.cs_pin(_host_cs(cs, n, n, p, i, x))
        B = rng.uniform(-1, 1, size=(n, k))
        pairs = ((cs.cs_lsolve, CO.lsolve), (cs.cs_ltsolve, CO.ltsolve)) if lower else \
                ((cs.cs_usolve, CO.usolve), (cs.cs_utsolve, CO.utsolve))
        for fn, ofn in pairs:
            dB = cs.dvec(B)
            assert fn(T, dB) is True
            X = dB.numpy()
            for r in (0, 63, 64, 129):
                assert X[:, r].tobytes() == ofn(n, p, i, x, B[:, r]).tobytes(), (fn.__name__, r)
