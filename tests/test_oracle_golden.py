"""Pin the CPU oracle (oracle/csparse_oracle.py) against vectors produced by the
unmodified reference (oracle/gen_golden.py) and the known answers of the
reference's own tests (csparse_test.py, transcribed in SURVEY.md section 4)."""
import hashlib

import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack, same_csc

SMALL = ["t1", "bcsstk01", "west0067", "ash219", "fs_183_1", "ibm32a", "ibm32b", "lp_afiro"]
ALL = SMALL + ["bcsstk16", "mbeacxc"]


def sha(a, dt):
    return hashlib.sha256(np.asarray(a, dtype=dt).tobytes()).hexdigest()


def triplet(g):
    m, n, nzmax, nz = (int(v) for v in g["T_mn"])
    T = O.cs_spalloc(0, 0, 1, True, True)
    for i, j, x in zip(g["T_i"], g["T_j"], g["T_x"]):
        assert O.cs_entry(T, int(i), int(j), float(x))
    assert (T.m, T.n, T.nzmax, T.nz) == (m, n, nzmax, nz)
    return T


@pytest.mark.parametrize("name", ALL)
def test_compress_transpose_gaxpy_multiply(name, meta):
    g = golden(name)
    A = O.cs_compress(triplet(g))
    same_csc(A, g, "A")
    assert O.cs_norm(A) == meta[name]["normA"]
    AT = O.cs_transpose(A, True)
    same_csc(AT, g, "AT")
    assert O.cs_norm(AT) == meta[name]["normAT"]
    assert O.cs_transpose(A, False).x is None
    y = [float(v) for v in g["gaxpy_y0"]]
    assert O.cs_gaxpy(A, [float(v) for v in g["gaxpy_x"]], y) is True
    assert np.asarray(y).tobytes() == g["gaxpy_y"].tobytes()
    C = O.cs_multiply(A, AT)
    mm = meta[name]["AAT"]
    nnz = C.p[C.n]
    assert (C.m, C.n, nnz, C.nzmax, len(C.i)) == (mm["m"], mm["n"], mm["nnz"], mm["nzmax"], mm["leni"])
    assert sha(C.p, np.int64) == mm["sha_p"]
    assert sha(C.i[:nnz], np.int64) == mm["sha_i"]
    assert sha(C.x[:nnz], np.float64) == mm["sha_x"]
    assert O.cs_norm(C) == mm["norm"]
    if name in SMALL:
        same_csc(C, g, "AAT")
        same_csc(O.cs_multiply(O.cs_transpose(A, False), A), g, "ATA_pat")
    # D = A*A' + |A*A'|_1 * I, the Test1 assertion (csparse_test.py:262-266)
    Eye = O.cs_spalloc(C.m, C.m, C.m, True, False)
    Eye.p, Eye.i, Eye.x = list(range(C.m + 1)), list(range(C.m)), [1.0] * C.m
    D = O.cs_add(C, Eye, 1, O.cs_norm(C))
    assert D.p[D.n] == meta[name]["D"]["nnz"] and O.cs_norm(D) == meta[name]["D"]["norm"]


# known answers of csparse_test.py Test1 (:269-426): nnz(D), |D|_1 with the tests' own delta
TEST1 = {"t1": (16, 139.58), "bcsstk01": (764, 1.73403e19), "bcsstk16": (544856, 4.13336e19),
         "west0067": (1041, 61.0906), "ash219": (2205, 32.0), "fs_183_1": (19665, 2.80249e18),
         "ibm32a": (386, 70.0), "ibm32b": (373, 64.0), "lp_afiro": (153, 128.963),
         "mbeacxc": (157350, 19.6068)}   # csparse_test.py:381-394


@pytest.mark.parametrize("name", ALL)
def test_reference_test1_known_answers(name, meta):
    nnz, norm = TEST1[name]
    assert meta[name]["D"]["nnz"] == nnz
    assert meta[name]["D"]["norm"] == pytest.approx(norm, rel=1e-5)


def problem(name, g):
    A = O.cs_compress(triplet(g))
    O.cs_dupl(A)
    O.cs_dropzeros(A)
    O.cs_droptol(A, 1e-14)
    return A


def make_sym(A):
    AT = O.cs_transpose(A, True)
    O.cs_fkeep(AT, lambda i, j, a, o: i != j, None)
    return O.cs_add(A, AT, 1, 1)


@pytest.mark.parametrize("name", ALL)
def test_problem_pipeline_and_trisolves(name, meta):
    g = golden(name)
    A = problem(name, g)
    C = make_sym(A) if meta[name]["sym"] else A
    same_csc(C, g, "C")
    if "x_lsolve" in g:
        b = [float(v) for v in g["b"]]
        Lo, Up = unpack(O, g, "Lo"), unpack(O, g, "Up")
        for nm, fn, M in (("lsolve", O.cs_lsolve, Lo), ("ltsolve", O.cs_ltsolve, Lo),
                          ("usolve", O.cs_usolve, Up), ("utsolve", O.cs_utsolve, Up)):
            v = list(b)
            assert fn(M, v) is True
            assert np.asarray(v).tobytes() == g["x_" + nm].tobytes(), nm


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_lusol_against_reference(name, meta):
    g = golden(name)
    C = unpack(O, g, "C")
    tol = 0.001 if meta[name]["sym"] else 1.0
    b = [float(v) for v in g["b"]]
    # shipped loop reproduces the reference's factors and solution bit for bit
    S = O.cs_sqr(0, C, False)
    N = O.cs_lu(C, S, tol, shipped_quirk=True)
    same_csc(N.L, g, "refL")
    same_csc(N.U, g, "refU")
    assert N.pinv == [int(v) for v in g["ref_pinv"]]
    v = list(b)
    assert O.cs_lusol(0, C, v, tol, shipped_quirk=True)
    assert np.asarray(v).tobytes() == g["x_lusol"].tobytes()
    # the reference's own quirky L/U through the oracle's triangular solves
    w = [float(t) for t in g["ref_lu_pb"]]
    O.cs_lsolve(unpack(O, g, "refL"), w)
    assert np.asarray(w).tobytes() == g["ref_lu_y"].tobytes()
    O.cs_usolve(unpack(O, g, "refU"), w)
    assert np.asarray(w).tobytes() == g["ref_lu_x"].tobytes()
    # loop bounded as its comment says: same solution to rounding
    v2 = list(b)
    assert O.cs_lusol(0, C, v2, tol)
    np.testing.assert_allclose(v2, g["x_lusol"], rtol=1e-9, atol=0)


# csparse_test.py Test2 known answers (||x||_inf, absolute delta 1e-3)
TEST2 = {"t1": 2.4550, "bcsstk01": 0.0005, "west0067": 21.9478, "fs_183_1": 212022.2099}


@pytest.mark.parametrize("name", sorted(TEST2))
def test_reference_test2_known_answers(name, meta):
    assert meta[name]["lusol_norm_inf"] == pytest.approx(TEST2[name], abs=1e-3)


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_symbolic_pieces(name, meta):
    g = golden(name)
    C = unpack(O, g, "C")
    Cu = O.cs_symperm(C, None, False)
    same_csc(Cu, g, "symperm")
    parent = O.cs_etree(Cu, False)
    assert parent == [int(v) for v in g["etree"]]
    assert O.cs_post(parent, C.n) == [int(v) for v in g["post"]]
    if "ereach_top" in g:
        n = C.n
        w, s, tops, pats = [0] * n, [0] * n, [], []
        for k in range(n):
            top = O.cs_ereach(Cu, k, parent, s, 0, w)
            tops.append(top)
            pats.extend(s[top:n])
        assert tops == [int(v) for v in g["ereach_top"]]
        assert pats == [int(v) for v in g["ereach_pat"]]


def test_cholesky_restatement_bcsstk01(meta):
    """cs_chol/cs_schol/cs_cholsol do not run in the reference;
    pin the restatement by SURVEY 8c: lnz, structure, L L' = C, and agreement with
    the unmodified cs_lusol answer (csparse_test.py:505-516 expects 0.0005)."""
    g = golden("bcsstk01")
    C = unpack(O, g, "C")
    n = C.n
    assert (n, C.p[n]) == (48, 400)
    assert O.cs_norm(C) == pytest.approx(3.57094807469e9, rel=1e-11)
    S = O.cs_schol(0, C)
    assert S.lnz == 877 and S.cp[n] == 877
    N = O.cs_chol(C, S)
    L = N.L
    assert L.p == S.cp
    for j in range(n):
        col = L.i[L.p[j]:L.p[j + 1]]
        assert col[0] == j and col == sorted(col) and len(set(col)) == len(col)
    LLT = O.cs_multiply(L, O.cs_transpose(L, True))
    R = O.cs_add(LLT, C, 1, -1)
    assert O.cs_norm(R) / O.cs_norm(C) <= 1e-14
    b = [float(v) for v in g["b"]]
    x = list(b)
    assert O.cs_cholsol(0, C, x) is True
    ref = g["x_lusol"]
    assert max(abs(v) for v in x) == pytest.approx(0.0005, abs=1e-4)
    assert np.max(np.abs(np.asarray(x) - ref) / np.abs(ref)) < 1e-10
    # the explicit Test3 sequence (csparse_test.py:671-674)
    y = [0.0] * n
    O.cs_ipvec(S.pinv, b, y, n)
    O.cs_lsolve(L, y)
    O.cs_ltsolve(L, y)
    z = [0.0] * n
    O.cs_pvec(S.pinv, y, z, n)
    assert z == x


def test_cholesky_restatement_bcsstk16_symbolic():
    g = golden("bcsstk16")
    C = unpack(O, g, "C")
    assert C.p[C.n] == 290378
    assert O.cs_norm(C) == pytest.approx(7.008379365769155e9, rel=1e-12)
    assert O.cs_schol(0, C).lnz == 610800


def test_synthetic_edge_cases():
    g = golden("synthetic_20240601")
    for c, (m, n, nnz) in enumerate(g["cases"]):
        pre = "c%d_" % c
        A = unpack(O, g, pre + "A")
        AT = O.cs_transpose(A, True)
        same_csc(AT, g, pre + "AT")
        y = [float(v) for v in g[pre + "y0"]]
        O.cs_gaxpy(A, [float(v) for v in g[pre + "x"]], y)
        assert np.asarray(y).tobytes() == g[pre + "y"].tobytes()
        same_csc(O.cs_multiply(A, AT), g, pre + "AAT")
        same_csc(O.cs_multiply(AT, A), g, pre + "ATA")
        Ap = unpack(O, g, pre + "A")
        Ap.x = None
        same_csc(O.cs_multiply(Ap, AT), g, pre + "AAT_pat")
        if m != n:
            assert O.cs_multiply(A, A) is None
    c = [int(v) for v in g["cumsum_c"]]
    p = [0] * (len(c) + 1)
    assert O.cs_cumsum(p, c, len(c)) == int(g["cumsum_ret"][0])
    assert p == [int(v) for v in g["cumsum_p"]] and c == [int(v) for v in g["cumsum_c_out"]]
    perm = [int(v) for v in g["perm"]]
    b = [float(v) for v in g["perm_b"]]
    x = [0.0] * len(b)
    O.cs_ipvec(perm, b, x, len(b))
    assert x == [float(v) for v in g["ipvec"]]
    O.cs_pvec(perm, b, x, len(b))
    assert x == [float(v) for v in g["pvec"]]


def test_error_conventions():
    A = O.cs_spalloc(3, 3, 1, True, True)  # a triplet is not CSC
    assert O.cs_gaxpy(A, [0.0] * 3, [0.0] * 3) is False
    assert O.cs_gaxpy(None, [], []) is False
    assert O.cs_transpose(A, True) is None and O.cs_multiply(A, A) is None
    assert O.cs_lsolve(A, [0.0]) is False and O.cs_usolve(None, [0.0]) is False
    assert O.cs_cumsum(None, [1], 1) == -1
    assert O.cs_ipvec(None, None, [0.0], 1) is False
    L = O.cs_spalloc(1, 1, 1, True, False)
    L.p, L.i, L.x = [0, 1], [0], [0.0]
    with pytest.raises(ZeroDivisionError):
        O.cs_lsolve(L, [1.0])


# ---- the plain-C oracle must agree bit for bit with the pinned Python oracle ----
import c_oracle as CO  # noqa: E402


def _arr(A):
    nnz = A.p[A.n]
    return (np.asarray(A.p, dtype=np.int32), np.asarray(A.i[:nnz], dtype=np.int32),
            None if A.x is None else np.asarray(A.x[:nnz], dtype=np.float64))


@pytest.mark.parametrize("name", ALL)
def test_c_oracle_kernels(name, meta):
    g = golden(name)
    A = unpack(O, g, "A")
    Ap, Ai, Ax = _arr(A)
    y = CO.gaxpy(A.m, A.n, Ap, Ai, Ax, g["gaxpy_x"], g["gaxpy_y0"])
    assert y.tobytes() == g["gaxpy_y"].tobytes()
    Tp, Ti, Tx = CO.transpose(A.m, A.n, Ap, Ai, Ax)
    assert Tp.tolist() == g["AT_p"].tolist() and Ti.tolist() == g["AT_i"].tolist()
    assert Tx.tobytes() == g["AT_x"].tobytes()
    Cp, Ci, Cx = CO.multiply(A.m, A.n, A.m, Ap, Ai, Ax, Tp, Ti, Tx)
    mm = meta[name]["AAT"]
    assert sha(Cp, np.int64) == mm["sha_p"] and sha(Ci, np.int64) == mm["sha_i"]
    assert sha(Cx, np.float64) == mm["sha_x"]
    Pp, Pi, Px = CO.multiply(A.m, A.n, A.m, Ap, Ai, None, Tp, Ti, Tx)
    assert Px is None and Pi.tolist() == Ci.tolist()
    if "x_lsolve" in g:
        Lp, Li, Lx = _arr(unpack(O, g, "Lo"))
        Up, Ui, Ux = _arr(unpack(O, g, "Up"))
        n = A.n
        assert CO.lsolve(n, Lp, Li, Lx, g["b"]).tobytes() == g["x_lsolve"].tobytes()
        assert CO.ltsolve(n, Lp, Li, Lx, g["b"]).tobytes() == g["x_ltsolve"].tobytes()
        assert CO.usolve(n, Up, Ui, Ux, g["b"]).tobytes() == g["x_usolve"].tobytes()
        assert CO.utsolve(n, Up, Ui, Ux, g["b"]).tobytes() == g["x_utsolve"].tobytes()
    if "refL" in g:
        Lp, Li, Lx = _arr(unpack(O, g, "refL"))
        Up, Ui, Ux = _arr(unpack(O, g, "refU"))
        yv = CO.lsolve(A.n, Lp, Li, Lx, g["ref_lu_pb"])
        assert yv.tobytes() == g["ref_lu_y"].tobytes()
        assert CO.usolve(A.n, Up, Ui, Ux, yv).tobytes() == g["ref_lu_x"].tobytes()


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_c_oracle_cholesky(name):
    g = golden(name)
    C = unpack(O, g, "C")
    n = C.n
    Cp, Ci, Cx = _arr(C)
    parent, cp = CO.schol(n, Cp, Ci)
    S = O.cs_schol(0, C)
    assert parent.tolist() == S.parent and cp.tolist() == S.cp
    if name == "bcsstk01":
        N = O.cs_chol(C, S)
        Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
        assert Lp.tolist() == N.L.p and Li.tolist() == N.L.i
        assert Lx.tobytes() == np.asarray(N.L.x).tobytes()
        bad = Cx.copy()
        bad[[p for j in range(n) for p in range(Cp[j], Cp[j + 1]) if Ci[p] == j][3]] = -1.0
        assert CO.chol(n, Cp, Ci, bad, parent, cp) is None
        Cbad = unpack(O, g, "C")
        Cbad.x = [float(v) for v in bad]
        assert O.cs_chol(Cbad, S) is None
    else:
        Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
        assert Lp[n] == 610800
        b = g["b"]
        x = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b))
        # csparse_test.py:528-533: ||x||_inf = 1.9998 +- 1e-3
        assert np.max(np.abs(x)) == pytest.approx(1.9998, abs=1e-3)
        r = CO.gaxpy(n, n, Cp, Ci, Cx, x, -b)
        assert np.max(np.abs(r)) / (O.cs_norm(C) * np.max(np.abs(x)) + np.max(np.abs(b))) < 1e-14


def test_c_oracle_perm_and_zero_pivot():
    g = golden("synthetic_20240601")
    assert CO.ipvec(g["perm"], g["perm_b"]).tolist() == g["ipvec"].tolist()
    assert CO.pvec(g["perm"], g["perm_b"]).tolist() == g["pvec"].tolist()
    assert CO.ipvec(None, g["perm_b"]).tolist() == g["perm_b"].tolist()
    with pytest.raises(ZeroDivisionError):
        CO.lsolve(1, [0, 1], [0], [0.0], [1.0])
    with pytest.raises(ZeroDivisionError):
        CO.usolve(1, [0, 1], [0], [-0.0], [1.0])


def test_config2_fixture_pins_both_oracles():
    """BASELINE config 2 (cs_gaxpy on the symmetric-expanded bcsstk16): the unmodified reference's y."""
    g, v = golden("bcsstk16"), golden("config2_bcsstk16")
    C = unpack(O, g, "C")
    assert C.p[C.n] == 290378
    y = v["y0"].tolist()
    assert O.cs_gaxpy(C, v["x"].tolist(), y)
    assert np.asarray(y).tobytes() == v["y"].tobytes()
    p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
    assert CO.gaxpy(4884, 4884, p, i, x, v["x"], v["y0"]).tobytes() == v["y"].tobytes()
