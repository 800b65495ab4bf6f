"""Parity of the device triangular solves and vector permutations against the
oracle and the golden vectors (needs an MI355X)."""
import numpy as np
import pytest

import c_oracle as CO
import csparse_oracle as O
import synth
from conftest import golden, unpack
from test_gpu_parity import ALL, _host_cs, cs  # noqa: F401  (cs is the module fixture)

pytestmark = pytest.mark.gpu


def _solvers(mod):
    return {"lsolve": mod.cs_lsolve, "ltsolve": mod.cs_ltsolve, "usolve": mod.cs_usolve, "utsolve": mod.cs_utsolve}


@pytest.mark.parametrize("name", ALL)
def test_trisolves_bit_exact_on_reference_matrices(cs, name):
    g = golden(name)
    if "x_lsolve" not in g:
        pytest.skip("no triangular fixture (rectangular or zero diagonal)")
    Lo, Up = unpack(cs, g, "Lo"), unpack(cs, g, "Up")
    for nm, fn in _solvers(cs).items():
        x = g["b"].tolist()
        alias = x
        assert fn(Lo if nm.startswith("l") else Up, x) is True
        assert alias is x and np.asarray(x).tobytes() == g["x_" + nm].tobytes(), nm


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_trisolves_on_the_reference_lu_factors(cs, name):
    """L and U as the reference's own cs_lu emits them: unsorted columns, duplicate rows,
    explicit zeros.  cs_lusol's solve sequence must come out bit-identical."""
    g = golden(name)
    L, U = unpack(cs, g, "refL"), unpack(cs, g, "refU")
    n = L.n
    b = g["b"].tolist()
    x = [0.0] * n
    assert cs.cs_ipvec(g["ref_pinv"].tolist(), b, x, n)
    assert x == g["ref_lu_pb"].tolist()
    assert cs.cs_lsolve(L, x) and np.asarray(x).tobytes() == g["ref_lu_y"].tobytes()
    assert cs.cs_usolve(U, x) and np.asarray(x).tobytes() == g["ref_lu_x"].tobytes()
    out = [0.0] * n
    assert cs.cs_ipvec(None, x, out, n) and np.asarray(out).tobytes() == g["x_lusol"].tobytes()


def test_trisolve_many_rhs_and_device_permutation(cs):
    g = golden("bcsstk16")
    n, k = 4884, 7
    Lo, Up = cs.cs_pin(unpack(cs, g, "Lo")), cs.cs_pin(unpack(cs, g, "Up"))
    B = synth.rhs(n, k, 3)
    oLo, oUp = unpack(O, g, "Lo"), unpack(O, g, "Up")
    for nm in ("lsolve", "ltsolve", "usolve", "utsolve"):
        X = cs.dvec(B)
        assert _solvers(cs)[nm](Lo if nm.startswith("l") else Up, X) is True
        got = X.numpy()
        for r in range(k):
            ref = B[:, r].tolist()
            _solvers(O)[nm](oLo if nm.startswith("l") else oUp, ref)
            assert got[:, r].tobytes() == np.asarray(ref).tobytes(), (nm, r)
    rng = np.random.default_rng(3)
    perm = rng.permutation(n).astype(np.int32)
    db, dx = cs.dvec(B), cs.dvec(n, k)
    assert cs.cs_ipvec(perm.tolist(), db, dx, n)
    ref = np.empty_like(B)
    ref[perm] = B
    assert dx.numpy().tobytes() == ref.tobytes()
    assert cs.cs_pvec(perm.tolist(), db, dx, n)
    assert dx.numpy().tobytes() == B[perm].tobytes()
    assert cs.cs_pvec(None, db, dx, n) and dx.numpy().tobytes() == B.tobytes()


def test_trisolve_zero_pivot_and_malformed(cs):
    L = cs.cs_spalloc(3, 3, 4, True, False)
    L.p, L.i, L.x = [0, 2, 3, 4], [0, 2, 1, 2], [2.0, 1.0, 0.0, 4.0]
    with pytest.raises(ZeroDivisionError):
        cs.cs_lsolve(L, [1.0, 1.0, 1.0])
    with pytest.raises(ZeroDivisionError):
        cs.cs_ltsolve(L, [1.0, 1.0, 1.0])
    # an entry above the diagonal in a non-first slot: the reference's push loop still runs
    # (it rewrites an already final unknown); the device must reproduce that, not "fix" it
    M = cs.cs_spalloc(4, 4, 7, True, False)
    M.p, M.i, M.x = [0, 2, 4, 6, 7], [0, 3, 1, 0, 2, 3, 3], [2.0, 0.5, 4.0, 0.25, 5.0, 1.5, 8.0]
    Mo = O.cs_spalloc(4, 4, 7, True, False)
    Mo.p, Mo.i, Mo.x = list(M.p), list(M.i), list(M.x)
    for nm in ("lsolve", "ltsolve"):
        x, ref = [1.0, 2.0, 3.0, 4.0], [1.0, 2.0, 3.0, 4.0]
        assert _solvers(cs)[nm](M, x) and _solvers(O)[nm](Mo, ref)
        assert x == ref, nm
    # a column with no entries has no diagonal
    E = cs.cs_spalloc(2, 2, 1, True, False)
    E.p, E.i, E.x = [0, 1, 1], [0], [1.0]
    with pytest.raises((ValueError, IndexError)):
        cs.cs_lsolve(E, [1.0, 1.0])


def _random_lower(n, per_col, seed):
    """Sparse lower-triangular CSC with the diagonal first in each column, rows unsorted."""
    rng = np.random.default_rng(seed)
    cols_i, cols_x, Ap = [], [], [0]
    for j in range(n):
        cnt = min(per_col, n - 1 - j)
        rows = (j + 1 + rng.choice(n - 1 - j, size=cnt, replace=False)) if cnt > 0 else np.empty(0, dtype=np.int64)
        cols_i.append(np.concatenate([[j], rows]))
        cols_x.append(np.concatenate([[2.0 + rng.random()], rng.uniform(-0.2, 0.2, cnt)]))
        Ap.append(Ap[-1] + cnt + 1)
    return (np.asarray(Ap, dtype=np.int32), np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x))


def test_trisolve_medium_random_against_c_oracle(cs):
    n = 6000
    Lp, Li, Lx = _random_lower(n, 6, 9)
    L = cs.cs_pin(_host_cs(cs, n, n, Lp, Li, Lx))
    b = synth.vec(n, 4, 0.5, 1.5)
    for nm, ref in (("lsolve", CO.lsolve(n, Lp, Li, Lx, b)), ("ltsolve", CO.ltsolve(n, Lp, Li, Lx, b))):
        x = cs.dvec(b)
        assert _solvers(cs)[nm](L, x)
        assert x.numpy().tobytes() == ref.tobytes(), nm
    # U: transposing the diagonal-first lower factor gives rows ascending with the diagonal last
    Up, Ui, Ux = CO.transpose(n, n, Lp, Li, Lx)
    U = cs.cs_pin(_host_cs(cs, n, n, Up, Ui, Ux))
    for nm, ref in (("usolve", CO.usolve(n, Up, Ui, Ux, b)), ("utsolve", CO.utsolve(n, Up, Ui, Ux, b))):
        x = cs.dvec(b)
        assert _solvers(cs)[nm](U, x)
        assert x.numpy().tobytes() == ref.tobytes(), nm


def _band_triangles(gx, gy, seed):
    """Lower triangle L (diagonal first, rows ascending) of a gx x gy natural-order grid pattern's band, FULL inside the
    band of half-width gx like a Cholesky factor of it, well conditioned; and U = L' (diagonal last)."""
    import scipy.sparse as sp
    n = gx * gy
    rng = np.random.default_rng(seed)
    offs = list(range(0, gx + 1))
    diags = [4.0 + rng.random(n)] + [0.5 * (rng.random(n - d) - 0.5) / gx for d in offs[1:]]
    L = sp.diags(diags, [-d for d in offs], shape=(n, n), format="csc")
    L.sort_indices()
    U = L.T.tocsc()
    U.sort_indices()
    return n, L, U


@pytest.mark.parametrize("gx,gy", [(200, 90), (60, 300)])
def test_chain_like_banded_systems_too_big_for_lds_run_on_a_window_of_x(cs, gx, gy):
    """n > 15 360 (x does not fit LDS), every column handing on to its neighbour, band half-width gx: the column
    loops on a circular window of x (k_tri_wcolumns for L / U, k_tri_wcolchain for L' / U'), 1 024 threads and twelve
    rounds fetched ahead for the wide band, 256 and two for the narrow one.  Same operations in the same order as
    cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve (csparse.py:1330-1365, 2368-2385, 2460-2475): bit-identical,
    for one and for several right-hand sides."""
    n, L, U = _band_triangles(gx, gy, 11)
    assert n > 15360
    mats = {}
    for nm, M in (("L", L), ("U", U)):
        A = cs.cs_spalloc(n, n, M.nnz, True, False)
        A.p, A.i, A.x = M.indptr.tolist(), M.indices.tolist(), M.data.tolist()
        mats[nm] = (cs.cs_pin(A), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64))
    B = synth.rhs(n, 3, 5)
    refs = {"lsolve": CO.lsolve, "ltsolve": CO.ltsolve, "usolve": CO.usolve, "utsolve": CO.utsolve}
    for nm, fn in _solvers(cs).items():
        A, p, i, x = mats["L" if nm.startswith("l") else "U"]
        X1 = cs.dvec(B[:, 0].copy())
        assert fn(A, X1) is True
        assert X1.numpy().tobytes() == refs[nm](n, p, i, x, B[:, 0]).tobytes(), nm
        X = cs.dvec(B)
        assert fn(A, X) is True
        got = X.numpy()
        for r in range(3):
            assert got[:, r].tobytes() == refs[nm](n, p, i, x, B[:, r]).tobytes(), (nm, r)


def test_cholsol_on_a_wide_band_factor_both_orders(cs):
    """cs_cholsol's solve phase on the factor of a natural-order grid Laplacian (n = 18 000, band 200): the default
    order reproduces cs_lsolve + cs_ltsolve bit for bit; the rounding-equal order runs L' in push form on the rows
    of L (k_tri_wcolumns on the forward plan's gather arrays) and agrees to 1e-12."""
    from test_gpu_cholesky import _grid_laplacian
    n, p, i, x = _grid_laplacian(200, 90)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    cs.cs_pin(A)
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    b = synth.rhs(n, 2, 9)
    ref = [CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b[:, r])) for r in range(2)]
    for exact in (True, False):
        F = cs.cholsol_factor(A, 0, exact=exact)
        X = cs.dvec(b)
        assert F.solve(X) is True
        got = X.numpy()
        for r in range(2):
            if exact:
                assert got[:, r].tobytes() == ref[r].tobytes()
            else:
                assert np.max(np.abs(got[:, r] - ref[r])) / np.max(np.abs(ref[r])) < 1e-12


@pytest.mark.parametrize("k", [1, 3, 20, 70])
def test_long_rows_on_a_bushy_tree_take_a_wave_per_row(cs, k):
    """A 200 x 200 grid Laplacian factored in the order-1 (nested dissection) ordering: n = 40 000, a bushy tree, rows
    of 35 terms on average and separator rows of thousands, hundreds of narrow levels at the top.  Up to 4 right-hand
    sides: a wave per row, products through LDS to lane 0 (k_tri_level_rows); 16 or more: a wave per (row, 64
    right-hand sides), terms handed round by v_readlane (k_tri_level_rows64); runs of narrow levels in two phases
    (k_tri_run_prefix_rows / k_tri_run_prefix64, then the level walkers).  5..15 right-hand sides: a thread per (row,
    right-hand side).  Same subtractions in the same order as cs_lsolve / cs_ltsolve (csparse.py:1330-1365):
    bit-identical, also with a partly filled last block of 64."""
    from test_gpu_cholesky import _grid_laplacian
    n, p, i, x = _grid_laplacian(200, 200)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    cs.cs_pin(A)
    S = cs.cs_schol(1, A)
    N = cs.cs_chol(A, S)
    L = N.L
    lnz = L.p[n]
    assert lnz >= 24 * n                                  # long rows on average: the wave-per-row kernels are chosen
    Lp, Li, Lx = np.asarray(L.p, np.int32), np.asarray(L.i[:lnz], np.int32), np.asarray(L.x[:lnz], np.float64)
    cs.cs_pin(L)
    B = synth.rhs(n, k, 21)
    for nm, ref in (("lsolve", CO.lsolve), ("ltsolve", CO.ltsolve)):
        X = cs.dvec(B if k > 1 else B[:, 0].copy())
        assert _solvers(cs)[nm](L, X) is True
        got = X.numpy().reshape(n, k)
        for r in sorted({0, 1 % k, (k - 6) % k, k - 1}):
            assert got[:, r].tobytes() == ref(n, Lp, Li, Lx, B[:, r]).tobytes(), (nm, r)


def test_five_to_fifteen_right_hand_sides_on_the_same_factor(cs):
    """Between the two wave-per-row kernels: 9 right-hand sides go to the thread-per-(row, right-hand side) kernels."""
    test_long_rows_on_a_bushy_tree_take_a_wave_per_row(cs, 9)
