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


def _qr_both_ways(cs, n, Ap, Ai, Ax):
    """cs_qr through the device block path (pinned input) and through the host C++ code (csx_qr_host), same analysis"""
    import _csx
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_sqr(0, A, True)
    assert S is not None and S.m2 == n
    N = cs.cs_qr(A, S)
    assert N is not None and N.L._lazy                                  # it came from csx_qr_blocks
    parent, pinv, leftmost = _csx.i32(S.parent), _csx.i32(S.pinv), _csx.i32(S.leftmost)
    vcap, rcap = max(int(S.lnz), 1), max(int(S.unz), 1)
    Vp, Rp = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    Vi, Ri = np.zeros(vcap, np.int32), np.zeros(rcap, np.int32)
    Vx, Rx, beta = np.zeros(vcap), np.zeros(rcap), np.zeros(n)
    st = _csx.load().csx_qr_host(n, n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), None, _csx.pi(parent), _csx.pi(pinv),
                                 _csx.pi(leftmost), vcap, rcap, _csx.pi(Vp), _csx.pi(Vi), _csx.pd(Vx), _csx.pi(Rp),
                                 _csx.pi(Ri), _csx.pd(Rx), _csx.pd(beta))
    assert st == 0
    vnz, rnz = int(Vp[n]), int(Rp[n])
    assert N.L.p == Vp.tolist() and N.L.i[:vnz] == Vi[:vnz].tolist()
    assert N.U.p == Rp.tolist() and N.U.i[:rnz] == Ri[:rnz].tolist()
    assert np.asarray(N.L.x[:vnz]).tobytes() == Vx[:vnz].tobytes()
    assert np.asarray(N.U.x[:rnz]).tobytes() == Rx[:rnz].tobytes()
    assert np.asarray(N.B).tobytes() == beta.tobytes()
    return A


def test_qr_blocks_on_W_bit_identical_to_host_code(cs):
    """cs_qr for a batch of independent blocks on the device (csx_qr_blocks): W of BASELINE config 3, 200 blocks of
    west0067; V, R and beta equal the host C++ code's to the bit, and cs_qrsol on it solves the system."""
    import c_oracle as CO
    from test_gpu_configs import _w_matrix
    n, Ap, Ai, Ax = _w_matrix(200)
    A = _qr_both_ways(cs, n, Ap, Ai, Ax)
    b = 1.0 + np.arange(n) / n
    x = b.tolist()
    assert cs.cs_qrsol(0, A, x) is True
    res = CO.gaxpy(n, n, Ap, Ai, Ax, np.asarray(x), -b)
    norm1 = float(np.max(np.add.reduceat(np.abs(Ax), Ap[:-1])))
    assert np.max(np.abs(res)) <= 1e-11 * (norm1 * np.max(np.abs(x)) + np.max(np.abs(b)))
    # the device solve sequence (reflections applied level by level: those of a level touch disjoint rows) gives the
    # bits of the list-level cs_qrsol (reflections one after the other on the host) for every right-hand side
    rng = np.random.default_rng(5)
    B = np.column_stack([b] + [rng.uniform(-1, 1, n) for _ in range(2)])
    X = cs.qrsol_factor(A).solve(cs.dvec(B)).numpy().reshape(n, 3)
    assert X[:, 0].tobytes() == np.asarray(x).tobytes()
    for r in (1, 2):
        y = B[:, r].tolist()
        assert cs.cs_qrsol(0, A, y) is True and X[:, r].tobytes() == np.asarray(y).tobytes()


def test_qr_blocks_random_blocks(cs):
    """Blocks of mixed sizes, interleaved indices, structurally nonsingular (a cyclic off-diagonal keeps m2 == m)."""
    rng = np.random.default_rng(23)
    sizes = rng.integers(1, 50, size=130)
    n = int(sizes.sum())
    perm = rng.permutation(n)
    rows, cols, vals = [], [], []
    base = 0
    for m in sizes.tolist():
        mem = np.sort(perm[base:base + m])
        D = rng.uniform(-1, 1, (m, m)) * (rng.random((m, m)) < 0.25)
        D[np.arange(m), np.arange(m)] = rng.uniform(0.5, 1.5, m)
        for a in range(m):
            D[a, (a + 1) % m] += 0.7
        r, c = np.nonzero(D)
        rows.append(mem[r]); cols.append(mem[c]); vals.append(D[r, c])
        base += m
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((r, c))
    r, c, v = r[order], c[order], v[order]
    Ap = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=n))]).astype(np.int32)
    _qr_both_ways(cs, n, Ap, r.astype(np.int32), v.astype(np.float64))
