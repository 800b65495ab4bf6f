"""cs_spsolve / cs_reach / cs_dfs for all columns of B on the device (csx_spsolve) against the oracle called column
by column: the reach in xi[top..n-1] order and the solution beside it, bit for bit (needs an MI355X)."""
import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack
from test_gpu_parity import _host_cs, cs  # noqa: F401  (cs is the module fixture)
from test_gpu_fuzz import triangular

pytestmark = pytest.mark.gpu


def oracle_columns(G, B, pinv, lo, values=True):
    """the reference's loop: one cs_spsolve (or cs_reach) per column of B"""
    n = G.n
    p, idx, val = [0], [], []
    xi, x = [0] * (2 * n), [0.0] * n
    for k in range(B.n):
        top = O.cs_spsolve(G, B, k, xi, x, pinv, lo) if values else O.cs_reach(G, B, k, xi, pinv)
        idx += xi[top:n]
        if values:
            val += [x[j] for j in xi[top:n]]
        p.append(len(idx))
    return p, idx, val


def check(cs, G, B, oG, oB, pinv, lo):
    p, idx, val = oracle_columns(oG, oB, pinv, lo)
    X = cs.spsolve_columns(G, B, pinv, lo)
    assert (X.m, X.n) == (G.n, B.n)
    assert X.p == p and X.i[:p[-1]] == idx
    assert np.asarray(X.x[:p[-1]], dtype=np.float64).tobytes() == np.asarray(val, dtype=np.float64).tobytes()
    R = cs.reach_columns(G, B, pinv)
    assert R.p == p and R.i[:p[-1]] == idx and R.x is None


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_spsolve_on_the_reference_lu_factors(cs, name):
    """G = the L and U the unmodified reference's cs_lu produced (duplicates and explicit zeros included), B = A."""
    g = golden(name)
    L, U, A = unpack(cs, g, "refL"), unpack(cs, g, "refU"), unpack(cs, g, "A")
    oL, oU, oA = unpack(O, g, "refL"), unpack(O, g, "refU"), unpack(O, g, "A")
    pinv = g["ref_pinv"].tolist()
    check(cs, L, A, oL, oA, None, True)
    check(cs, U, A, oU, oA, None, False)
    check(cs, L, A, oL, oA, pinv, True)            # columns of L reached through a row permutation, as inside cs_lu
    partial = list(pinv)
    for j in range(0, len(partial), 5):
        partial[j] = -1                            # rows that are not pivotal yet: leaves of the search, skipped by the solve
    check(cs, L, A, oL, oA, partial, True)


@pytest.mark.parametrize("n,nb,mean_len,lower", [(1, 1, 0.0, True), (40, 200, 2.0, True), (300, 700, 3.0, False),
                                                 (1500, 90, 4.0, True)])
def test_spsolve_fuzz(cs, n, nb, mean_len, lower):
    rng = np.random.default_rng(n * 7 + nb)
    Gp, Gi, Gx = triangular(rng, n, mean_len, lower)
    lens = rng.integers(0, 4, size=nb)
    Bp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Bi = rng.integers(0, n, size=int(Bp[-1])).astype(np.int32)          # duplicates inside a column allowed
    Bx = rng.uniform(-1, 1, size=int(Bp[-1]))
    G, B = _host_cs(cs, n, n, Gp, Gi, Gx), _host_cs(cs, n, nb, Bp, Bi, Bx)
    oG, oB = _host_cs(O, n, n, Gp, Gi, Gx), _host_cs(O, n, nb, Bp, Bi, Bx)
    check(cs, G, B, oG, oB, None, lower)


def test_spsolve_more_columns_than_fit_in_flight(cs):
    """70 000 columns of B: more than the 65 536 lanes of one launch, so X is assembled from two pieces."""
    rng = np.random.default_rng(11)
    n, nb = 30, 70000
    Gp, Gi, Gx = triangular(rng, n, 3.0, True)
    Bp = np.arange(nb + 1, dtype=np.int32)
    Bi = rng.integers(0, n, size=nb).astype(np.int32)
    Bx = rng.uniform(-1, 1, size=nb)
    G, B = _host_cs(cs, n, n, Gp, Gi, Gx), _host_cs(cs, n, nb, Bp, Bi, Bx)
    oG, oB = _host_cs(O, n, n, Gp, Gi, Gx), _host_cs(O, n, nb, Bp, Bi, Bx)
    check(cs, G, B, oG, oB, None, True)


def test_spsolve_bad_input(cs):
    g = golden("t1")
    L, A = unpack(cs, g, "refL"), unpack(cs, g, "A")
    assert cs.spsolve_columns(None, A) is None
    wide = _host_cs(cs, L.n + 1, 1, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([1.0]))
    assert cs.spsolve_columns(L, wide) is None
    with pytest.raises(IndexError):
        cs.spsolve_columns(L, A, [L.n] * L.n, True)


def test_wrapped_arrays_with_bad_indices_are_refused(cs):
    """Arrays the library did not make (csx_csc_wrap) are checked on the device before the reach kernels index their
    per-lane work space with them: an out-of-range row index is CSX_EINVAL (IndexError in the reference), not a stray write."""
    import _csx
    lib = _csx.lib()
    n = 8
    Gp = np.arange(n + 1, dtype=np.int32)
    Gi = np.arange(n, dtype=np.int32)
    Gx = np.ones(n)
    hG = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Gp), _csx.pi(Gi), _csx.pd(Gx), hG))
    # B's arrays live in device vectors of ours and are wrapped; row index 8 is out of range
    for bad, expect in ((np.asarray([0, 3, 8], np.int32), _csx.EINVAL), (np.asarray([0, 3, 7], np.int32), _csx.OK)):
        hi = _csx.new_handle()
        _csx.check(lib.csx_ivec_upload(_csx.pi(bad), 3, hi))
        hp = _csx.new_handle()
        _csx.check(lib.csx_ivec_upload(_csx.pi(np.asarray([0, 3], np.int32)), 2, hp))
        hx = _csx.new_handle()
        _csx.check(lib.csx_vec_upload(_csx.pd(np.ones(3)), 3, hx))
        ptr = lambda h: (lambda p, l: (_csx.check(lib.csx_vec_ptr(h, p, l)), p.value)[1])(_csx.C.c_void_p(), _csx.C.c_int64())
        hB = _csx.new_handle()
        _csx.check(lib.csx_csc_wrap(n, 1, 3, ptr(hp), ptr(hi), ptr(hx), hB))
        hX = _csx.new_handle()
        assert lib.csx_spsolve(hG, hB, None, 1, 1, hX) == expect
        for h in (hB, hi, hp, hx) + ((hX,) if expect == _csx.OK else ()):
            _csx.free(h)
    _csx.free(hG)
