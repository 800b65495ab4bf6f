"""Host C++ steps inside libcsx (no GPU needed): cs_schol and cs_lu restated in C++
against the pinned oracle."""
import numpy as np
import pytest

import csparse_oracle as O
import synth
from conftest import golden, unpack


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_schol_host_matches_oracle(name):
    import csparse as cs
    g = golden(name)
    C = unpack(cs, g, "C")
    S = cs.cs_schol(0, C)
    R = O.cs_schol(0, unpack(O, g, "C"))
    assert S.parent == R.parent and S.cp == R.cp and S.lnz == R.lnz and S.pinv is None
    assert cs.cs_schol(2, C) is None  # only the natural order and the order-1 ordering are defined


def test_schol_host_block_diagonal():
    import csparse as cs
    Ap, Ai, Ax = synth.gspd(7, 8, 3)
    A = cs.cs_spalloc(56, 56, len(Ai), True, False)
    A.p, A.i, A.x = Ap.tolist(), Ai.tolist(), Ax.tolist()
    S = cs.cs_schol(0, A)
    assert S.lnz == 7 * 36
    assert S.parent == [(-1 if j % 8 == 7 else j + 1) for j in range(56)]


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_lu_host_matches_oracle(name, meta):
    import csparse as cs
    g = golden(name)
    tol = 0.001 if meta[name]["sym"] else 1.0
    C = unpack(cs, g, "C")
    N = cs.cs_lu(C, cs.cs_sqr(0, C, False), tol)
    Co = unpack(O, g, "C")
    R = O.cs_lu(Co, O.cs_sqr(0, Co, False), tol)
    assert N.pinv == R.pinv
    for got, ref in ((N.L, R.L), (N.U, R.U)):
        assert got.p == ref.p and got.i == ref.i
        assert np.asarray(got.x).tobytes() == np.asarray(ref.x).tobytes()
    # diagonal first in L (unit), last in U: what cs_lsolve / cs_usolve rely on
    n = C.n
    assert all(N.L.i[N.L.p[j]] == j and N.L.x[N.L.p[j]] == 1.0 for j in range(n))
    assert all(N.U.i[N.U.p[j + 1] - 1] == j for j in range(n))


def test_lu_host_singular_and_bad_input():
    import csparse as cs
    A = cs.cs_spalloc(2, 2, 2, True, False)
    A.p, A.i, A.x = [0, 1, 2], [0, 0], [1.0, 2.0]  # second row empty -> singular
    assert cs.cs_lu(A, cs.cs_sqr(0, A, False), 1.0) is None
    assert cs.cs_lusol(0, A, [1.0, 1.0], 1.0) is False
    assert cs.cs_lu(A, None, 1.0) is None and cs.cs_sqr(4, A, False) is None and cs.cs_sqr(-1, A, True) is None
    T = cs.cs_spalloc(2, 2, 2, True, True)
    assert cs.cs_lusol(0, T, [1.0, 1.0], 1.0) is False and cs.cs_lusol(0, A, None, 1.0) is False



@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1", "bcsstk16"])
def test_sqr_qr_host_matches_unmodified_reference(name, meta):
    """cs_sqr(0, C, True) in host C++ (csx_sqr_host) against what the unmodified reference computes for the square
    problem matrices of its own tests (tests/golden/sqr_qr.npz, oracle/gen_golden.py sqr): column elimination tree,
    column counts of R, row permutation, leftmost columns, sizes."""
    import csparse as cs
    g = golden(name)
    q = golden("sqr_qr")
    C = unpack(cs, g, "C")
    S = cs.cs_sqr(0, C, True)
    mm = meta["sqr_qr"][name]
    assert (S.m2, S.lnz, S.unz) == (mm["m2"], mm["lnz"], mm["unz"])
    assert S.parent == q[name + "_parent"].tolist() and S.cp == q[name + "_cp"].tolist()
    assert S.pinv[:S.m2] == q[name + "_pinv"][:S.m2].tolist()            # beyond m2 the array is unused work space
    assert S.leftmost == q[name + "_leftmost"].tolist()
    assert S.q is None


def test_sqr_qr_host_rectangular_and_rank_deficient():
    """m > n with an empty row and a column no row starts in: fictitious rows are numbered from m upwards and rows
    without a pivot from n upwards (CSparse's numbering; the reference's port collides there)."""
    import csparse as cs
    #      c0 c1 c2
    # r0 [  x  .  . ]
    # r1 [  x  x  . ]
    # r2 [  .  .  . ]
    # r3 [  x  .  x ]
    A = cs.cs_spalloc(4, 3, 5, True, False)
    A.p, A.i, A.x = [0, 3, 4, 5], [0, 1, 3, 1, 3], [1.0, 2.0, 3.0, 4.0, 5.0]
    S = cs.cs_sqr(0, A, True)
    assert S.leftmost == [0, 0, -1, 0]
    assert S.parent == [1, 2, -1]
    assert S.m2 == 4 and sorted(S.pinv[:4]) == [0, 1, 2, 3] and S.pinv[2] == 3   # the empty row goes last
    assert S.lnz == 3 + 2 + 1 and S.unz == 3 + 2 + 1


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "ash219", "fs_183_1", "ibm32a", "ibm32b", "lp_afiro", "bcsstk16"])
def test_counts_ata_matches_unmodified_reference(name, meta):
    """cs_counts(C, parent, post, True) (csparse.py:703-764; host C++, csx_counts_host) against the UNMODIFIED reference's
    answer for the problem matrix of every matrix of its tests, rectangular ones included (tests/golden/counts_ata.npz,
    oracle/gen_golden.py counts); the tree and the postorder fed to it are the reference's, and the product's own
    cs_etree / cs_post must reproduce them."""
    import csparse as cs
    g, q = golden(name), golden("counts_ata")
    C = unpack(cs, g, "C")
    parent, post, want = (q[name + "_" + k].tolist() for k in ("parent", "post", "count"))
    assert (C.m, C.n) == (meta["counts_ata"][name]["m"], meta["counts_ata"][name]["n"])
    assert cs.cs_counts(C, parent, post, True) == want
    assert sum(want) == meta["counts_ata"][name]["total"]
    assert cs.cs_etree(C, True) == parent and cs.cs_post(parent, C.n) == post
    if C.m == C.n and name != "bcsstk16":      # cs_sqr's counts of R are the same numbers (csparse.py:2206-2208)
        assert cs.cs_sqr(0, C, True).cp[:C.n] == want
    # bad input: None, like the reference (csparse.py:712-713)
    assert cs.cs_counts(None, parent, post, True) is None and cs.cs_counts(C, None, post, True) is None
    assert cs.cs_counts(C, parent, None, True) is None


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_counts_of_a_symmetric_matrix_are_cs_schols(name):
    """ata False does not run in the reference: pinned against the oracle's restatement and against what cs_schol
    builds from the same counts (cp = their cumulative sum, csparse.py:2069-2071; lnz = 877 / 610 800)."""
    import csparse as cs
    g = golden(name)
    C, Co = unpack(cs, g, "C"), unpack(O, g, "C")
    Cu = O.cs_symperm(Co, None, False)                    # cs_schol analyses the upper triangle (csparse.py:2063)
    parent = O.cs_etree(Cu, False)
    post = O.cs_post(parent, C.n)
    U = cs.cs_spalloc(C.n, C.n, max(len(Cu.i), 1), False, False)
    U.p, U.i, U.x = list(Cu.p), list(Cu.i), None
    got = cs.cs_counts(U, parent, post, False)
    assert got == O.cs_counts(Cu, parent, post, False)
    assert np.cumsum([0] + got).tolist() == cs.cs_schol(0, C).cp
    assert sum(got) == {"bcsstk01": 877, "bcsstk16": 610800}[name]
    # the full symmetric matrix gives the same counts: entries below the diagonal fail cs_leaf's i <= j test
    assert cs.cs_counts(C, parent, post, False) == got
    assert cs.cs_counts(unpack(cs, golden("ash219"), "C"), parent, post, False) is None      # not square
