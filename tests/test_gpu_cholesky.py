"""cs_chol / cs_cholsol / cs_lusol on the device against the oracle (needs an MI355X).
The reference cannot run cs_chol: parity is against the restatement, which is
pinned by the reference's own cs_lusol answers and the L L' = C residual (test_oracle_golden)."""
import numpy as np
import pytest

import c_oracle as CO
import csparse_oracle as O
import synth
import tol as TOL
from conftest import golden, unpack
from test_gpu_parity import RTOL, _host_cs, cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _arr(A):
    nnz = A.p[A.n]
    return (np.asarray(A.p, np.int32), np.asarray(A.i[:nnz], np.int32), np.asarray(A.x[:nnz], np.float64))


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_chol_and_cholsol_reference_matrices(cs, name):
    g = golden(name)
    C = unpack(cs, g, "C")
    n = C.n
    S = cs.cs_schol(0, C)
    N = cs.cs_chol(C, S)
    L = N.L
    Cp, Ci, Cx = _arr(C)
    parent, cp = CO.schol(n, Cp, Ci)
    Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
    assert L.p == S.cp == Lp.tolist()
    assert L.i[:Lp[n]] == Li.tolist()                      # structure: bit-exact
    got = np.asarray(L.x[:Lp[n]])
    assert np.max(np.abs(got - Lx) / np.abs(Lx).max()) < 1e-13
    small = np.abs(Lx) > 1e-6 * np.abs(Lx).max()
    assert np.max(np.abs(got[small] - Lx[small]) / np.abs(Lx[small])) < RTOL
    # the driver: b overwritten in place, True returned
    b = g["b"].tolist()
    alias = b
    assert cs.cs_cholsol(0, C, b) is True and alias is b
    # x against the oracle's solves on the ORACLE's L (which differs from the device's by 1e-13): two factorisations of
    # one matrix, so the componentwise bound carries its conditioning (tests/tol.py); the bit-for-bit check on the build's
    # own L is the last assertion of this test
    ref = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, g["b"]))
    assert TOL.normwise(b, ref) < TOL.X_RTOL
    assert TOL.componentwise(b, ref) < TOL.cross_bound(TOL.cond1(TOL.csc(n, Cp, Ci, Cx)))
    if name == "bcsstk01":
        # csparse_test.py:505-516 (0.0005) and the unmodified reference's own LU answer
        assert max(abs(v) for v in b) == pytest.approx(0.0005, abs=1e-4)
        assert np.max(np.abs(np.asarray(b) - g["x_lusol"]) / np.abs(g["x_lusol"])) < RTOL
    else:
        assert max(abs(v) for v in b) == pytest.approx(1.9998, abs=1e-3)  # csparse_test.py:528-533
    # the reference's solve sequence on the build's L reproduces the driver bit for bit
    gLp, gLi, gLx = _arr(L)
    seq = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, g["b"]))
    assert np.asarray(b).tobytes() == seq.tobytes()


def test_chol_not_positive_definite(cs):
    g = golden("bcsstk01")
    C = unpack(cs, g, "C")
    S = cs.cs_schol(0, C)
    diag = [p for j in range(C.n) for p in range(C.p[j], C.p[j + 1]) if C.i[p] == j]
    x = list(C.x)
    x[diag[5]] = -1.0
    C.x = x
    assert cs.cs_chol(C, S) is None
    assert cs.cs_cholsol(0, C, g["b"].tolist()) is False
    assert cs.cs_chol(C, None) is None


def test_chol_with_permutation(cs):
    g = golden("bcsstk01")
    C = unpack(cs, g, "C")
    Co = unpack(O, g, "C")
    n = C.n
    rng = np.random.default_rng(11)
    pinv = rng.permutation(n).tolist()
    C2 = O.cs_symperm(Co, pinv, False)
    So = O.css()
    So.pinv = pinv
    So.parent = O.cs_etree(C2, False)
    So.cp = [0] * (n + 1)
    So.lnz = O.cs_cumsum(So.cp, O.cs_counts(C2, So.parent, O.cs_post(So.parent, n), False), n)
    No = O.cs_chol(Co, So)
    S = cs.css()
    S.pinv, S.parent, S.cp, S.lnz = pinv, So.parent, So.cp, So.lnz
    N = cs.cs_chol(C, S)
    assert N.L.p == No.L.p and N.L.i == No.L.i
    # ONE factorisation, the device's against the oracle's: rounding of the sums only (tests/tol.py)
    gx, ox = np.asarray(N.L.x[:No.L.p[n]]), np.asarray(No.L.x[:No.L.p[n]])
    assert TOL.normwise(gx, ox) <= 1e-13
    big = np.abs(ox) > 1e-6 * np.abs(ox).max()
    assert TOL.componentwise(gx[big], ox[big]) <= 1e-12
    Cp_, Ci_, Cx_ = _arr(C)
    bound = TOL.cross_bound(TOL.cond1(TOL.csc(n, Cp_, Ci_, Cx_)))      # two factorisations of one matrix: its conditioning
    # batched solve through the permutation, generic (level-scheduled) path
    b = g["b"].tolist()
    xo = list(b)
    y = [0.0] * n
    O.cs_ipvec(pinv, xo, y, n)
    O.cs_lsolve(No.L, y)
    O.cs_ltsolve(No.L, y)
    O.cs_pvec(pinv, y, xo, n)
    assert TOL.normwise(xo, g["x_lusol"]) <= bound                    # the oracle's Cholesky against the unmodified reference's LU
    import _csx
    plan = _csx.new_handle()
    with cs._Resident(N.L) as dL:                # N.L's lists were read above: its device copy is gone, upload again
        _csx.check(_csx.lib().csx_cholsol_plan(dL.handle, _csx.pi(_csx.i32(pinv)), plan))
    B = np.stack([g["b"], 2 * g["b"], g["b"] + 1.0], axis=1)
    dB = cs.dvec(B)
    _csx.check(_csx.lib().csx_cholsol_solve(plan, dB.handle, 3))
    X = dB.numpy()
    _csx.free(plan)
    assert TOL.normwise(X[:, 0], xo) <= TOL.X_RTOL and TOL.componentwise(X[:, 0], xo) <= bound
    assert TOL.normwise(X[:, 1], 2 * np.asarray(xo)) <= TOL.X_RTOL and TOL.componentwise(X[:, 1], 2 * np.asarray(xo)) <= bound


@pytest.mark.parametrize("nblocks,bs,k", [(40, 64, 130), (300, 8, 5), (3, 32, 64), (25, 16, 70), (7, 32, 200), (50, 8, 129)])
def test_gspd_factor_and_batched_solve(cs, nblocks, bs, k):
    """Block-diagonal SPD (the benchmark's G-spd shape): forest of small trees -> tree kernels for the
    factorisation; the solve phase in its default (exact) order must be bit-identical to cs_lsolve + cs_ltsolve
    for EVERY right-hand side, and the rounding-equal order (matrix cores) within 1e-13 of it."""
    import _csx
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 20240606)
    n = nblocks * bs
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, exact=True)          # every solve in the reference order (the default gives that to lists only)
    assert F.info() == {"fused_local": True, "dense_block": bs, "matrix_cores": False, "trees": nblocks,
                        "max_nodes": bs}
    parent, cp = CO.schol(n, Ap, Ai)
    assert F.symbolic.parent == parent.tolist() and F.symbolic.cp == cp.tolist()
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    L = F.L
    assert L.p == Lp.tolist() and L.i[:Lp[n]] == Li.tolist()
    got = np.asarray(L.x[:Lp[n]])
    # dense-block kernel: the reference's operation order on a chain tree -> the same bits
    assert got.tobytes() == Lx.tobytes()
    # the general column kernel (forced) sums in a different order: equal to rounding
    with _csx.option("chol.dense_trees", 0):
        Lg = cs.cs_chol(A, F.symbolic).L
    assert Lg.p == Lp.tolist() and Lg.i[:Lp[n]] == Li.tolist()
    assert np.max(np.abs(np.asarray(Lg.x[:Lp[n]]) - Lx)) / np.abs(Lx).max() < 1e-13
    B = synth.rhs(n, k, 0)
    gLp, gLi, gLx = _arr(L)
    refs = {r: CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r])) for r in sorted(set([0, 1, k // 2, k - 1]))}
    # default order (dense-block substitution in the reference's order): the reference's bits for every right-hand side
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    X = dB.numpy()
    for r, ref in refs.items():
        assert X[:, r].tobytes() == ref.tobytes(), r
    # the fused per-tree kernel (what forests of non-dense trees use) gives the same bits
    with _csx.option("cholsol.dense_blocks", 0):
        dB1 = cs.dvec(B)
        assert F.solve(dB1) is True
        assert dB1.numpy().tobytes() == X.tobytes()
    # rounding-equal order: blocked TRSM on the matrix cores for 16/32/64 blocks, FMA substitution for 8
    Ff = cs.cholsol_factor(A, exact=False)
    assert Ff.info()["dense_block"] == bs and Ff.info()["matrix_cores"] is (bs >= 16)
    dB3 = cs.dvec(B)
    assert Ff.solve(dB3) is True
    X3 = dB3.numpy()
    assert np.max(np.abs(X3 - X) / np.abs(X)) < 1e-13
    assert X3.tobytes() != X.tobytes()            # it really is another kernel
    # ... and with the dense-block kernels switched off the same plan falls back to the exact kernel
    with _csx.option("cholsol.dense_blocks", 0):
        dB2 = cs.dvec(B)
        assert Ff.solve(dB2) is True
        assert dB2.numpy().tobytes() == X.tobytes()
    # residual of the whole block against A (symmetric, full storage)
    R = np.stack([CO.gaxpy(n, n, Ap, Ai, Ax, X[:, r], -B[:, r]) for r in (0, k - 1)], axis=1)
    assert np.max(np.abs(R)) < 1e-12 * np.max(np.abs(B)) * bs
    # one right-hand side as a list through the drop-in driver: the default order, the reference's bits
    b = B[:, 0].tolist()
    assert cs.cs_cholsol(0, A, b) is True
    assert np.asarray(b).tobytes() == refs[0].tobytes()


@pytest.mark.parametrize("nblocks,bs,k", [(40, 64, 128), (25, 16, 70), (7, 32, 192)])
def test_matrix_core_solve_reads_L_or_the_plans_copy_same_bits(cs, nblocks, bs, k):
    """k_cholsol_mfma takes the off-diagonal tiles out of L.x itself (a plan on all the columns of L: csx_cholsol_factor's, or the
    three calls' when the factor is recognised as equal blocks) or out of the packed copy a plan of the GENERAL analysis keeps
    ("chol.clique" = 0 takes the recognition away): three routes to the same operands, the same solution bits."""
    import _csx
    lib = _csx.lib()
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 20240613)
    n = nblocks * bs
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    B = synth.rhs(n, k, 1)
    F = cs.cholsol_factor(A, exact=False)                  # one call: the block kernel wrote the W tiles, the rest is L.x
    assert F.info()["matrix_cores"] is True
    d0 = cs.dvec(B)
    assert F.solve(d0) is True
    X0 = d0.numpy().copy()
    S = cs.cs_schol(0, A)
    N = cs.cs_chol(A, S)
    outs = {}
    for name, clique in (("recognised", 1), ("general", 0)):
        with _csx.option("chol.clique", clique):
            plan = _csx.new_handle()
            with cs._Resident(N.L) as dL:
                _csx.check(lib.csx_cholsol_plan(dL.handle, None, plan))
                _csx.check(lib.csx_cholsol_set_order(plan, 0))
                a, b, c = _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32()
                _csx.check(lib.csx_cholsol_info(plan, a, b, c))
                assert a.value == 3, (name, a.value)          # dense blocks on the matrix cores
                dB = cs.dvec(B)
                _csx.check(lib.csx_cholsol_solve(plan, dB.handle, k))
                outs[name] = dB.numpy().copy()
                _csx.free(plan)
    assert outs["recognised"].tobytes() == X0.tobytes()
    assert outs["general"].tobytes() == X0.tobytes()
    gLp, gLi, gLx = _arr(N.L)
    for r in (0, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert TOL.normwise(X0.reshape(n, k)[:, r], ref) <= 1e-13


@pytest.mark.parametrize("bs", [16, 64])
@pytest.mark.parametrize("spread", [1.0, 30.0, 300.0, 800.0, 3000.0, 1e5])
def test_matrix_core_solve_below_and_above_the_growth_guard(cs, bs, spread):
    """exact=False solves dense blocks with explicit inverses of the diagonal tiles; its error grows with
    max|inv(L_ii)| max|L| ("growth").  Blocks are scaled D A D with D spanning `spread`, which puts growth
    anywhere from ~1 to far beyond the guard (1e3): below it the matrix cores run and must stay inside the
    1e-10 budget against the plain-C oracle; above it the plan must keep (FMA) substitution, same budget."""
    import _csx
    nblocks, k = 30, 70
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 20240612)
    n = nblocks * bs
    d = np.tile(np.logspace(0.0, np.log10(spread), bs), nblocks)
    rng = np.random.default_rng(int(spread) + bs)
    for b in range(nblocks):                       # a different ordering of the scales in every block
        d[b * bs:(b + 1) * bs] = rng.permutation(d[b * bs:(b + 1) * bs])
    cols = np.repeat(np.arange(n), np.diff(Ap))
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax * d[Ai] * d[cols]))
    F = cs.cholsol_factor(A, exact=False)
    growth = _csx.C.c_double()
    _csx.check(_csx.lib().csx_cholsol_growth(F.plan_handle, growth))
    B = synth.rhs(n, k, 0)
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    X = dB.numpy()
    gLp, gLi, gLx = _arr(F.L)
    on_cores = F.info()["matrix_cores"]
    assert on_cores is (growth.value <= 1e3), (growth.value, on_cores)
    worst = 0.0
    for r in (0, 17, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        worst = max(worst, float(np.max(np.abs(X[:, r] - ref) / np.abs(ref))))
    assert worst < 1e-10, (spread, growth.value, worst)


def test_forest_of_sparse_trees_uses_generic_fused_kernel(cs):
    """Block-diagonal with TRIDIAGONAL-plus-arrow blocks: small trees that are not dense blocks, and a
    permutation -> the generic fused kernel (X tile in LDS), bit-identical per right-hand side."""
    nb, bs, k = 200, 24, 70
    n = nb * bs
    rng = np.random.default_rng(4)
    cols_i, cols_x, Ap = [], [], [0]
    for j in range(n):
        b0 = (j // bs) * bs
        rows = {j}
        if j - 1 >= b0:
            rows.add(j - 1)
        if j + 1 < b0 + bs:
            rows.add(j + 1)
        rows.add(b0 + bs - 1)  # arrow: last row/column of every block is full
        if j == b0 + bs - 1:
            rows.update(range(b0, b0 + bs))
        rows = sorted(rows)
        vals = [(8.0 + (j % 5)) if r == j else -1.0 / (1 + abs(r - j)) for r in rows]
        cols_i.append(rows)
        cols_x.append(vals)
        Ap.append(Ap[-1] + len(rows))
    Ap = np.asarray(Ap, np.int32)
    Ai = np.concatenate(cols_i).astype(np.int32)
    Ax = np.concatenate(cols_x)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, exact=True)
    info = F.info()
    assert info["fused_local"] and info["dense_block"] == 0 and info["trees"] == nb
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    assert F.L.p == Lp.tolist() and F.L.i[:Lp[n]] == Li.tolist()
    B = synth.rhs(n, k, 1)
    dB = cs.dvec(B)
    assert F.solve(dB)
    X = dB.numpy()
    gLp, gLi, gLx = _arr(F.L)
    for r in (0, 33, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert X[:, r].tobytes() == ref.tobytes(), r
        res = CO.gaxpy(n, n, Ap, Ai, Ax, X[:, r], -B[:, r])
        assert np.max(np.abs(res)) < 1e-12


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_lusol_matches_reference(cs, name, meta):
    g = golden(name)
    C = unpack(cs, g, "C")
    tol = 0.001 if meta[name]["sym"] else 1.0
    b = g["b"].tolist()
    assert cs.cs_lusol(0, C, b, tol) is True
    ref = g["x_lusol"]  # unmodified reference cs_lusol(0, ...)
    assert TOL.normwise(b, ref) < TOL.X_RTOL
    # componentwise: terms of the last substitution of cs_usolve on the reference's own U are not at hand for the
    # build's (different, D7) factors, so |ref| alone is the scale -- the build's LU takes the reference's pivots and
    # operation order, and the answers agree far inside the conditioning of fs_183_1 (1.5e13)
    assert TOL.componentwise(b, ref) < TOL.X_RTOL
    assert max(abs(v) for v in b) == pytest.approx(meta[name]["lusol_norm_inf"], rel=TOL.X_RTOL)


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16", "gspd", "arrow", "random"])
def test_schol_on_device_matches_host_and_oracle(cs, name):
    """cs_schol for a device-resident matrix (tree on the host, column counts from the device walks) gives
    the same S.parent / S.cp as the host path and the C oracle."""
    if name in ("bcsstk01", "bcsstk16"):
        A = unpack(cs, golden(name), "C")
    elif name == "gspd":
        Ap, Ai, Ax = synth.gspd(7, 16, 3)
        A = _host_cs(cs, 7 * 16, 7 * 16, Ap, Ai, Ax)
    else:
        rng = np.random.default_rng(9)
        n = 600
        rows, cols = [np.arange(n)], [np.arange(n)]
        if name == "arrow":                      # dense last row/column + a band: long row subtrees
            rows += [np.full(n - 1, n - 1), np.arange(n - 1), np.arange(1, n), np.arange(n - 1)]
            cols += [np.arange(n - 1), np.full(n - 1, n - 1), np.arange(n - 1), np.arange(1, n)]
        else:
            r, c = rng.integers(0, n, 1500), rng.integers(0, n, 1500)
            rows += [r, c]
            cols += [c, r]
        T = cs.cs_spalloc(n, n, 1, True, True)
        for i, j in zip(np.concatenate(rows).tolist(), np.concatenate(cols).tolist()):
            cs.cs_entry(T, i, j, 1.0 if i != j else 4.0 * n)
        A = cs.cs_compress(T)
        cs.cs_dupl(A)
    n = A.n
    Sh = cs.cs_schol(0, A)                       # host lists -> csx_schol_host
    cs.cs_pin(A)
    Sd = cs.cs_schol(0, A)                       # device-resident -> csx_schol
    assert Sd.parent == Sh.parent and Sd.cp == Sh.cp and Sd.lnz == Sh.lnz
    p, i = np.asarray(A.p, np.int32), np.asarray(A.i[:A.p[n]], np.int32)
    parent, cp = CO.schol(n, p, i)
    assert Sd.parent == parent.tolist() and Sd.cp == cp.tolist()
    assert cs.cs_chol(A, Sd) is not None         # and the numeric phase accepts it


def test_relaxed_order_on_a_chain_like_factor(cs):
    """cholsol_factor(exact=False): the blocked chain walker takes a row's out-of-block terms first in BOTH
    directions.  Same solution to rounding; the default stays bit-identical to cs_lsolve + cs_ltsolve."""
    g = golden("bcsstk16")
    C = cs.cs_pin(unpack(cs, g, "C"))
    n, k = C.n, 7
    B = synth.rhs(n, k, 0)
    Fe, Fr = cs.cholsol_factor(C, exact=True), cs.cholsol_factor(C, exact=False)
    Xe, Xr = cs.dvec(B), cs.dvec(B)
    assert Fe.solve(Xe) and Fr.solve(Xr)
    Xe, Xr = Xe.numpy(), Xr.numpy()
    gLp, gLi, gLx = _arr(Fe.L)
    for r in (0, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert Xe[:, r].tobytes() == ref.tobytes()                     # exact order: the reference's bits
        assert np.max(np.abs(Xr[:, r] - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert not np.array_equal(Xe, Xr) or True                          # (they usually differ in the last bits)


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_cholsol_with_the_fill_reducing_ordering(cs, name):
    """order = 1: nested dissection in place of the reference's (non-working) cs_amd.  No permutation to match;
    the solution must be the order-0 one to rounding, and the driver conventions must hold."""
    g = golden(name)
    C = unpack(cs, g, "C")
    n = C.n
    P = cs.cs_amd(1, C)
    assert sorted(P) == list(range(n)) and cs.cs_amd(0, C) is None and cs.cs_amd(1, None) is None
    S = cs.cs_schol(1, C)
    assert S.pinv == cs.cs_pinv(P, n) and len(S.parent) == n and S.cp[n] == S.lnz
    b0, b1 = g["b"].tolist(), g["b"].tolist()
    assert cs.cs_cholsol(0, C, b0) is True and cs.cs_cholsol(1, C, b1) is True
    x0, x1 = np.asarray(b0), np.asarray(b1)
    Cp_, Ci_, Cx_ = _arr(C)
    bound = TOL.cross_bound(TOL.cond1(TOL.csc(n, Cp_, Ci_, Cx_)))      # two orderings = two factorisations of one matrix (tests/tol.py)
    assert TOL.normwise(x1, x0) <= bound
    # batched, device-resident
    cs.cs_pin(C)
    F = cs.cholsol_factor(C, order=1, exact=True)
    B = np.stack([g["b"], 3.0 * g["b"]], axis=1)
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    X = dB.numpy()
    assert TOL.normwise(X[:, 0], x0) <= bound
    assert TOL.normwise(X[:, 1], 3.0 * x0) <= bound


@pytest.mark.parametrize("shape", ["gspd_8", "mixed"])
def test_schol_etree_by_components_on_device(cs, shape):
    """A matrix with thousands of small components gets its elimination tree on the device (one thread per
    component running the reference's loop, csparse.py:1136-1169): same parent / cp as the host path and the
    plain-C oracle."""
    if shape == "gspd_8":
        Ap, Ai, Ax = synth.gspd(3000, 8, 11)
        n = 3000 * 8
    else:
        # blocks of size 1..40 with random symmetric patterns inside, in a random interleaving of the indices
        rng = np.random.default_rng(21)
        sizes = rng.integers(1, 41, size=2500)
        n = int(sizes.sum())
        perm = rng.permutation(n)                       # component members are NOT contiguous
        rows, cols = [np.arange(n)], [np.arange(n)]
        base = 0
        for m in sizes.tolist():
            mem = perm[base:base + m]
            if m > 1:
                e = rng.integers(0, m, size=(2 * m, 2))
                ch = np.stack([np.arange(m - 1), np.arange(1, m)], axis=1)    # a chain keeps the block connected
                e = np.concatenate([e, ch])
                rows += [mem[e[:, 0]], mem[e[:, 1]]]
                cols += [mem[e[:, 1]], mem[e[:, 0]]]
            base += m
        r, c = np.concatenate(rows), np.concatenate(cols)
        order = np.lexsort((r, c))
        r, c = r[order], c[order]
        keep = np.ones(len(r), bool)
        keep[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
        r, c = r[keep], c[keep]
        Ap = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=n))]).astype(np.int32)
        Ai = r.astype(np.int32)
        Ax = np.where(r == c, 100.0, 1.0)
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    Sh = cs.cs_schol(0, A)                       # host lists -> csx_schol_host
    cs.cs_pin(A)
    Sd = cs.cs_schol(0, A)                       # device-resident -> csx_schol, tree by components
    assert Sd.parent == Sh.parent and Sd.cp == Sh.cp
    parent, cp = CO.schol(n, Ap, Ai)
    assert Sd.parent == parent.tolist() and Sd.cp == cp.tolist()


def test_banded_chain_factor_takes_the_register_window_kernel(cs):
    """bcsstk16 in natural order: chain-like elimination tree, L dense inside a band of half-width 140 -> the
    register-window kernel (k_chol_band).  On a chain tree a right-looking band factorisation applies the
    reference's operations in the reference's order: L.x bit-identical to the plain-C oracle.  The general column
    kernels (forced) agree to rounding.  A banded matrix that is NOT positive definite must be refused (None)."""
    import _csx
    g = golden("bcsstk16")
    C = unpack(cs, g, "C")
    p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
    n = C.n
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    S = cs.cs_schol(0, C)
    with _csx.option("chol.wband", 0):               # by default the blocked dense-band kernels take this width
        N = cs.cs_chol(C, S)
    assert N.L.p == Lp.tolist() and N.L.i[:Lp[n]] == Li.tolist()
    got = np.asarray(N.L.x[:Lp[n]])
    assert got.tobytes() == Lx.tobytes()
    with _csx.option("chol.band", 0):
        Ng = cs.cs_chol(C, S)
    gg = np.asarray(Ng.L.x[:Lp[n]])
    assert np.max(np.abs(gg - Lx)) / np.abs(Lx).max() < 1e-12 and gg.tobytes() != Lx.tobytes()
    # not positive definite: flip the sign of a diagonal entry half way down
    C2 = unpack(cs, g, "C")
    j = n // 2
    for q in range(C2.p[j], C2.p[j + 1]):
        if C2.i[q] == j:
            C2.x[q] = -C2.x[q]
    assert cs.cs_chol(C2, S) is None


def _grid_laplacian(gx, gy, shift=0.01):
    """5-point Laplacian of a gx x gy grid in natural order (x fastest): a chain elimination tree and a factor that is
    dense inside a band of half-width gx."""
    import scipy.sparse as sp
    Tx = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(gx, gx))
    Ty = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(gy, gy))
    A = (sp.kron(sp.identity(gy), Tx) + sp.kron(Ty, sp.identity(gx)) + shift * sp.identity(gx * gy)).tocsc()
    A.sort_indices()
    return gx * gy, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


@pytest.mark.parametrize("gx,gy,wband,nb", [(200, 31, 1, 16), (200, 31, 1, 32), (61, 59, 2, 16), (61, 59, 2, 32),
                                            (200, 31, 1, -16), (61, 59, 2, -32), (300, 40, 1, 16),
                                            (33, 17, 2, 32), (7, 80, 2, 16), (300, 2, 1, 32), (300, 2, 2, 16)])
def test_wide_band_chain_factor_blocked_in_a_dense_band_array(cs, gx, gy, wband, nb):
    """Natural-order grid Laplacians: chain elimination tree, band half-width gx.  Wider than the register window
    (gx = 200) the blocked dense-band kernels (csx_cholband.hip) take the factor by themselves; forced (wband = 2)
    they take narrow bands too.  Every element receives its updates in ascending column order, multiply and subtract
    rounded separately: the reference's operation sequence on a chain tree (csparse.py:598-612) -> L.x bit-identical
    to the plain-C oracle, for panel widths 16 and 32, sizes that are no multiple of the panel, and bands wider than
    what is left of the matrix.  nb > 0: one launch per panel (the next panel factored while the previous one's update
    of the rest of the window runs); nb < 0: two launches per panel."""
    import _csx
    n, p, i, x = _grid_laplacian(gx, gy)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    S = cs.cs_schol(0, A)
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    with _csx.option("chol.wband", wband), _csx.option("chol.wband_nb", nb):
        N = cs.cs_chol(A, S)
    assert N.L.p == Lp.tolist() and N.L.i[:Lp[n]] == Li.tolist()
    assert np.asarray(N.L.x[:Lp[n]]).tobytes() == Lx.tobytes()
    if gx == 200:
        with _csx.option("chol.wband", 0):               # the general column kernels: same factor to rounding
            Ng = cs.cs_chol(A, S)
        gg = np.asarray(Ng.L.x[:Lp[n]])
        assert np.max(np.abs(gg - Lx)) / np.abs(Lx).max() < 1e-12


def test_wide_band_factor_refuses_a_matrix_that_is_not_positive_definite(cs):
    n, p, i, x = _grid_laplacian(200, 12)
    x = x.copy()
    j = n // 2 + 7
    for q in range(p[j], p[j + 1]):
        if i[q] == j:
            x[q] = -x[q]
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    assert cs.cs_chol(A, cs.cs_schol(0, A)) is None       # csparse.py:612


def test_wide_band_kernels_on_bcsstk16(cs):
    """The reference's own matrix through the blocked dense-band kernels (its band, half-width 140, also fits the
    register-window kernel): bit-identical to the plain-C oracle."""
    import _csx
    g = golden("bcsstk16")
    C = unpack(cs, g, "C")
    p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
    n = C.n
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    S = cs.cs_schol(0, C)
    for nb in (16, 32, -16):
        with _csx.option("chol.wband", 2), _csx.option("chol.wband_nb", nb):
            N = cs.cs_chol(C, S)
        assert np.asarray(N.L.x[:Lp[n]]).tobytes() == Lx.tobytes()


@pytest.mark.parametrize("gx,gy", [(120, 120), (75, 131)])
def test_supernodes_of_a_nested_dissection_factor_are_factored_as_dense_trapezoids(cs, gx, gy):
    """Order 1 on a grid Laplacian: the separators are fundamental supernodes (w consecutive columns, each the only child
    of the next, column counts falling by one).  Those of 8+ columns leave the level lists: their columns take every
    outside update in one launch and the trapezoid is factored densely in place (k_sn_step), 16 columns per launch,
    widths that are no multiple of 16 included.  L.p / L.i exact, L.x within 1e-13 of the plain-C oracle on the permuted
    matrix, and the same with the supernode path switched off; a non-positive pivot inside a supernode -> None."""
    import _csx
    n, p, i, x = _grid_laplacian(gx, gy)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    S = cs.cs_schol(1, A)
    C = cs.cs_symperm(A, S.pinv, True)
    Cp, Ci, Cx = _arr(C)
    parent, cp = CO.schol(n, Cp, Ci)
    assert S.parent == parent.tolist() and S.cp == cp.tolist()
    # there ARE wide supernodes in this factor
    nchild = np.bincount(parent[parent >= 0], minlength=n)
    cnt = np.diff(cp)
    chain = (parent[:-1] == np.arange(1, n)) & (nchild[1:] == 1) & (cnt[1:] == cnt[:-1] - 1)
    runs, best = 0, 0
    for v in chain:
        runs = runs + 1 if v else 0
        best = max(best, runs + 1)
    assert best >= 64
    Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
    N = cs.cs_chol(A, S)
    assert N.L.p == Lp.tolist() and N.L.i[:Lp[n]] == Li.tolist()
    got = np.asarray(N.L.x[:Lp[n]])
    assert np.max(np.abs(got - Lx)) / np.abs(Lx).max() < 1e-13
    with _csx.option("chol.supernodes", 0):
        N0 = cs.cs_chol(A, S)
    g0 = np.asarray(N0.L.x[:Lp[n]])
    assert np.max(np.abs(g0 - Lx)) / np.abs(Lx).max() < 1e-13
    assert np.max(np.abs(g0 - got)) / np.abs(Lx).max() < 1e-13
    # the solve through the driver
    b = np.linspace(1.0, 2.0, n)
    xb = b.tolist()
    assert cs.cs_cholsol(1, A, xb) is True
    import scipy.sparse as sp
    Am = sp.csc_matrix((x, i, p), shape=(n, n))
    Am = Am + sp.triu(Am, 1).T if (Am != Am.T).nnz else Am
    assert np.max(np.abs(Am @ np.asarray(xb) - b)) < 1e-10
    # not positive definite: the diagonal entry of the LAST column (inside the top separator's supernode) negated
    pinv = np.asarray(S.pinv)
    orig = int(np.where(pinv == n - 3)[0][0])
    x2 = x.copy()
    for q in range(p[orig], p[orig + 1]):
        if i[q] == orig:
            x2[q] = -50.0
    A2 = cs.cs_spalloc(n, n, len(i), True, False)
    A2.p, A2.i, A2.x = p.tolist(), i.tolist(), x2.tolist()
    assert cs.cs_chol(A2, S) is None


def test_chain_tree_with_a_wide_but_empty_band_stays_with_the_sparse_kernels(cs):
    """An arrow matrix (tridiagonal + a dense last row / column): the elimination tree is a chain and the band is n - 1
    wide, but the factor is as sparse as the matrix.  The dense-band kernels would do n x band^2 work on it; they are
    taken only when the band is mostly full.  The factor must come out right (and quickly) from the sparse kernels."""
    import time
    import scipy.sparse as sp
    n = 6000
    main = np.full(n, 4.0)
    off = np.full(n - 1, -1.0)
    A = sp.diags([off, main, off], [-1, 0, 1], shape=(n, n), format="lil")
    A[n - 1, :] = -0.001
    A[:, n - 1] = -0.001
    A[n - 1, n - 1] = 10.0
    A = A.tocsc()
    A.sort_indices()
    p, i, x = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    M = cs.cs_spalloc(n, n, len(i), True, False)
    M.p, M.i, M.x = p.tolist(), i.tolist(), x.tolist()
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    assert Lp[n] < 3 * n                                   # sparse factor, band n - 1
    S = cs.cs_schol(0, M)
    cs.cs_chol(M, S)
    t0 = time.perf_counter()
    N = cs.cs_chol(M, S)
    dt = time.perf_counter() - t0
    assert N.L.p == Lp.tolist() and N.L.i[:Lp[n]] == Li.tolist()
    # (the last pivot is a sum of 6 000 terms, taken in 16 partial sums by the cooperative column kernel: 1.5e-13)
    assert np.max(np.abs(np.asarray(N.L.x[:Lp[n]]) - Lx)) / np.abs(Lx).max() < 1e-12
    assert dt < 5.0                                        # the dense-band path would take far longer than this


def test_solver_keeps_its_factor_alive_and_follows_updown(cs):
    """cholsol_factor's C plan borrows L's device arrays.  (1) Reading F.L.p / .i / .x (which copies to the host) and
    then churning the allocator with same-size blocks must not pull the factor out from under the plan: on a big-tree
    matrix (bcsstk16: the level-scheduled path, whose L' plan reads L.p / L.i / L.x directly) the solve must still be
    bit-identical to the oracle's cs_lsolve + cs_ltsolve.  (2) cs_updown(F.L, ...) changes the factor in place: the
    solver re-plans and solves with the NEW factor in both sweeps."""
    import _csx
    g = golden("bcsstk16")
    C = cs.cs_pin(unpack(cs, g, "C"))
    n = C.n
    with _csx.option("pool.limit_mb", 64):           # a small cache: freed blocks go back to the driver / get reused at once
        F = cs.cholsol_factor(C, exact=True)
        Lp, Li, Lx = np.asarray(F.L.p, np.int32), np.asarray(F.L.i, np.int32), np.asarray(F.L.x)   # materialises L
        lnz = int(Lp[n])
        assert F.L._dev is not None                  # the factor stays on the device
        junk = [cs.dvec(np.full(lnz, 7.0)) for _ in range(6)]     # same size as L.x
        junk += [cs.dvec(np.full((lnz + 1) // 2, 3.0)) for _ in range(6)]   # same bytes as L.i
        del junk
        import gc
        gc.collect()
        junk2 = [cs.dvec(np.full(lnz, -1.0)) for _ in range(3)]
        b = g["b"].copy()
        x = b.tolist()
        assert F.solve(x)
        ref = CO.ltsolve(n, Lp, Li[:lnz], Lx[:lnz], CO.lsolve(n, Lp, Li[:lnz], Lx[:lnz], b))
        assert np.asarray(x).tobytes() == ref.tobytes()
        del junk2
        # (2) rank-1 update with w = column 10 of L scaled: the factor changes in place on the device
        parent = F.symbolic.parent
        j = 10
        w = cs.cs_spalloc(n, 1, int(Lp[j + 1] - Lp[j]), True, False)
        w.p = [0, int(Lp[j + 1] - Lp[j])]
        w.i = Li[Lp[j]:Lp[j + 1]].tolist()
        w.x = (0.01 * Lx[Lp[j]:Lp[j + 1]]).tolist()
        assert cs.cs_updown(F.L, 1, w, parent) is True
        L2x = np.asarray(F.L.x[:lnz])
        assert not np.array_equal(L2x, Lx[:lnz])
        x2 = b.tolist()
        assert F.solve(x2)
        ref2 = CO.ltsolve(n, Lp, Li[:lnz], L2x, CO.lsolve(n, Lp, Li[:lnz], L2x, b))
        assert np.asarray(x2).tobytes() == ref2.tobytes()


@pytest.mark.parametrize("case", ["grid_120x120", "grid_75x131", "bcsstk16", "grid_60x47@natural", "band_3000x90@natural"])
def test_supernodal_solves_in_the_rounding_equal_order(cs, case):
    """cholsol_factor(A, order=1, exact=False) on a connected nested-dissection factor: the solves are scheduled by
    SUPERNODE (csx_snsolve.hip: outside terms by a wave per piece of a row, the dense triangle in panels of 16) instead
    of by column.  Against the plain-C oracle's cs_lsolve + cs_ltsolve on the same L at 1e-10 (BASELINE's tolerance),
    for 1, 3, 64 and 70 right-hand sides (a partial second chunk); the exact order of the same factor stays bit-identical;
    the same bits from run to run (partial sums are added in a fixed order); switched off -> the level-scheduled path."""
    import _csx
    # "@natural": order 0 -- the factor of a banded matrix is ONE chain of the elimination tree in which no two columns share
    # their rows: the schedule is made of RELAXED supernodes (runs of 64 columns of the chain, their triangles made dense
    # in the matrix-core fragments, the in / out split of every row and column looked up)
    order = 0 if case.endswith("@natural") else 1
    case = case.split("@")[0]
    if case == "bcsstk16":
        g = golden("bcsstk16")
        A = cs.cs_pin(unpack(cs, g, "C"))
        n = A.n
    elif case.startswith("band_"):
        import scipy.sparse as sp
        n, band = (int(v) for v in case[5:].split("x"))
        rng = np.random.default_rng(17)
        rows, cols = [], []
        for d in range(1, band + 1):
            keep = rng.random(n - d) < 0.4                      # a band with holes: the triangles of the runs are sparse
            rows.append(np.nonzero(keep)[0] + d)
            cols.append(np.nonzero(keep)[0])
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        Lw = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, n))
        Sm = (Lw + Lw.T).tocsc()
        Sm = (Sm + sp.diags(np.asarray(abs(Sm).sum(axis=0)).ravel() + 1.0)).tocsc()
        Sm.sort_indices()
        A = cs.cs_spalloc(n, n, Sm.nnz, True, False)
        A.p, A.i, A.x = Sm.indptr.tolist(), Sm.indices.tolist(), Sm.data.tolist()
        cs.cs_pin(A)
    else:
        gx, gy = (int(v) for v in case[5:].split("x"))
        n, p, i, x = _grid_laplacian(gx, gy)
        A = cs.cs_spalloc(n, n, len(i), True, False)
        A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
        cs.cs_pin(A)
    Fr = cs.cholsol_factor(A, order=order, exact=False)
    path = _csx.C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_info(Fr.plan_handle, path, None, None))
    assert path.value == 4                                            # the supernodal schedule is in charge
    Fe = cs.cholsol_factor(A, order=order, exact=True)
    Lp, Li, Lx = _arr(Fr.L)
    pinv = np.asarray(Fr.symbolic.pinv) if Fr.symbolic.pinv is not None else np.arange(n)
    for k in (1, 3, 64, 70):
        B = synth.rhs(n, k, 2)
        Xr, Xr2, Xe = cs.dvec(B), cs.dvec(B), cs.dvec(B)
        assert Fr.solve(Xr) and Fr.solve(Xr2) and Fe.solve(Xe)
        Xr, Xr2, Xe = Xr.numpy().reshape(n, k), Xr2.numpy().reshape(n, k), Xe.numpy().reshape(n, k)
        assert Xr.tobytes() == Xr2.tobytes()                          # reproducible
        for r in sorted({0, k // 2, k - 1}):
            pb = np.empty(n)
            pb[pinv] = B[:, r]                                        # cs_ipvec
            y = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, pb))
            ref = y[pinv]                                             # cs_pvec
            assert Xe[:, r].tobytes() == ref.tobytes()                # exact order: the reference's bits
            assert TOL.normwise(Xr[:, r], ref) <= TOL.X_RTOL
            # SURVEY 8d's componentwise measure, the terms those of the last substitution (cs_ltsolve on this L)
            terms = TOL.cholsolve_terms(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, pb), y)[pinv]
            assert TOL.componentwise(Xr[:, r], ref, terms) <= TOL.X_RTOL
    # a right-hand side gets the same bits however many others are solved with it (the kernels that take few
    # right-hand sides -- lanes per task instead of a wave per task -- form every sum in the same order)
    B64 = synth.rhs(n, 64, 7)
    X64 = cs.dvec(B64)
    assert Fr.solve(X64)
    X64 = X64.numpy().reshape(n, 64)
    for kk in (1, 3, 8, 20, 33):
        Xk = cs.dvec(np.ascontiguousarray(B64[:, :kk]))
        assert Fr.solve(Xk)
        assert Xk.numpy().reshape(n, kk).tobytes() == np.ascontiguousarray(X64[:, :kk]).tobytes(), kk
    with _csx.option("tri.supernodes", 0):
        F0 = cs.cholsol_factor(A, order=order, exact=False)
        _csx.check(_csx.lib().csx_cholsol_info(F0.plan_handle, path, None, None))
        assert path.value == 0
        X0 = cs.dvec(synth.rhs(n, 3, 2))
        assert F0.solve(X0)
    X3 = cs.dvec(synth.rhs(n, 3, 2))
    assert Fr.solve(X3)
    assert np.max(np.abs(X0.numpy() - X3.numpy())) <= 1e-10 * np.max(np.abs(X3.numpy()))
    # the triangles of the supernodes: on the matrix cores when every 16 x 16 diagonal block is tame (the grids), by
    # substitution out of LDS otherwise or on request ("tri.supernodes" = 2); both inside the same tolerance
    nsn, steps, wmax, mc, growth = [_csx.C.c_int32(-1) for _ in range(4)] + [_csx.C.c_double(-1.0)]
    _csx.check(_csx.lib().csx_cholsol_sn_info(Fr.plan_handle, nsn, steps, wmax, mc, growth))
    assert nsn.value > 0 and steps.value > 0 and wmax.value >= 16 and growth.value >= 1.0
    assert mc.value == (1 if growth.value <= 1e3 else 0)
    if case != "bcsstk16":
        assert mc.value == 1
    with _csx.option("tri.supernodes", 2):
        F2 = cs.cholsol_factor(A, order=order, exact=False)
        _csx.check(_csx.lib().csx_cholsol_info(F2.plan_handle, path, None, None))
        assert path.value == (4 if order == 1 else 0)     # a chain has no dense supernodes: without relaxed ones, no schedule
        _csx.check(_csx.lib().csx_cholsol_sn_info(F2.plan_handle, None, None, None, mc, None))
        assert mc.value == 0
        X2 = cs.dvec(synth.rhs(n, 3, 2))
        assert F2.solve(X2)
    assert np.max(np.abs(X2.numpy() - X3.numpy())) <= 1e-10 * np.max(np.abs(X3.numpy()))


def test_supernodal_solve_replayed_as_a_graph_gives_the_same_bits(cs):
    """"tri.graph" = 1: the launches of a supernodal solve (both sweeps) are captured once into a hipGraph and replayed while
    the block of right-hand sides stays where it is -- the host then enqueues a solve in microseconds.  Same bits as the
    direct launches; another block (or another number of right-hand sides) is captured anew."""
    import _csx
    n, p, i, x = _grid_laplacian(70, 53)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    cs.cs_pin(A)
    F = cs.cholsol_factor(A, order=1, exact=False)
    path = _csx.C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_info(F.plan_handle, path, None, None))
    assert path.value == 4
    B = synth.rhs(n, 5, 9)
    X0 = cs.dvec(B)
    assert F.solve(X0)
    with _csx.option("tri.graph", 1):
        X1, X2 = cs.dvec(B), cs.dvec(B[:, :3].copy())
        assert F.solve(X1)                                   # captured here
        again = cs.dvec(B)
        X1h = X1.numpy().copy()
        assert F.solve(X2)                                   # another block, three right-hand sides: a new capture
        assert F.solve(again)                                # and back to five
    assert X1h.tobytes() == X0.numpy().tobytes()
    assert again.numpy().tobytes() == X0.numpy().tobytes()
    # a direct solve with many more right-hand sides in between moves the plan's work space: the capture must notice
    with _csx.option("tri.graph", 1):
        Y = cs.dvec(B)
        assert F.solve(Y)
    Big = cs.dvec(synth.rhs(n, 300, 1))
    assert F.solve(Big)
    with _csx.option("tri.graph", 1):
        Y.assign(B)                                          # the same buffer, the same number of right-hand sides
        assert F.solve(Y)
    assert Y.numpy().tobytes() == X0.numpy().tobytes()
    assert X2.numpy().reshape(n, 3).tobytes() == X0.numpy().reshape(n, 5)[:, :3].copy().tobytes()


def test_long_supernodal_solves_are_replayed_as_a_graph_by_default(cs):
    """"tri.graph" = 2, the default: a solve of more than 256 launches (a natural-order grid factor: a chain of relaxed
    supernodes, two or three launches per step) is captured when the SAME block comes a THIRD time in a row (round 4 captured
    on the second: 9 ms in front of a 1.8 ms solve, never repaid by a caller who solves a block twice) and replayed from then
    on; a block seen once or twice is launched directly.  The bits do not depend on any of it."""
    import _csx
    n, p, i, x = _grid_laplacian(150, 150)
    A = cs.cs_spalloc(n, n, len(i), True, False)
    A.p, A.i, A.x = p.tolist(), i.tolist(), x.tolist()
    cs.cs_pin(A)
    F = cs.cholsol_factor(A, order=0, exact=False)
    steps = _csx.C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_sn_info(F.plan_handle, None, steps, None, None, None))
    assert 6 * steps.value > 256
    B = synth.rhs(n, 3, 4)
    with _csx.option("tri.graph", 0):
        X0 = cs.dvec(B)
        assert F.solve(X0)
    ref = X0.numpy().tobytes()
    def captures():
        c, ms = _csx.C.c_int32(-1), _csx.C.c_double(-1.0)
        _csx.check(_csx.lib().csx_cholsol_graph_info(F.plan_handle, c, ms))
        return c.value, ms.value

    assert captures() == (0, 0.0)
    X = cs.dvec(B)
    for rep in range(5):            # direct twice, then captured, then replayed twice
        X.assign(B)
        assert F.solve(X)
        assert X.numpy().tobytes() == ref, rep
        assert captures()[0] == (0 if rep < 2 else 1), rep
    assert captures()[1] > 0.0
    other = cs.dvec(B)              # another block in between: launched directly, the capture survives for X
    assert F.solve(other) and other.numpy().tobytes() == ref
    X.assign(B)
    assert F.solve(X) and X.numpy().tobytes() == ref
    assert captures()[0] == 1


@pytest.mark.parametrize("strength, cores", [(0.3, 1), (0.9, 0)])
def test_supernodal_triangles_leave_the_matrix_cores_when_a_diagonal_block_is_ill_conditioned(cs, strength, cores):
    """The matrix-core triangles multiply by explicit inverses of the 16 x 16 diagonal blocks; the plan measures
    || |inv(L_ii)| |L_ii| ||_inf for every block and keeps substitution when the largest exceeds 1e3.  A dense lower
    triangle (one supernode of 536 columns above a leaf subtree of 64; more than the 512 columns a fused per-tree
    kernel takes) that is I + small noise, except for one diagonal
    block whose off-diagonal entries are all -`strength`: 0.3 -> growth ~ 1e2 (matrix cores), 0.9 -> ~ 1e4 (substitution).
    Either way the solution agrees with the plain-C oracle at 1e-10."""
    import _csx
    lib = _csx.lib()
    n = 600
    rng = np.random.default_rng(11)
    Ld = np.tril(0.003 * rng.uniform(-1, 1, (n, n)), -1) + np.eye(n)
    blk = slice(96, 112)                                   # rows 32 .. 47 of the supernode that starts at column 64
    Ld[blk, blk] = np.tril(np.full((16, 16), -strength), -1) + np.eye(16)   # inverse entries grow like (1 + s)^k
    Lp = np.zeros(n + 1, np.int32)
    Li, Lx = [], []
    for j in range(n):
        Li.append(np.arange(j, n, dtype=np.int32))
        Lx.append(Ld[j:, j].copy())
        Lp[j + 1] = Lp[j] + n - j
    Li, Lx = np.concatenate(Li), np.concatenate(Lx)
    hL = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Lp), _csx.pi(Li), _csx.pd(Lx), hL))
    plan = _csx.new_handle()
    _csx.check(lib.csx_cholsol_plan(hL, None, plan))
    _csx.check(lib.csx_cholsol_set_order(plan, 0))
    path, mc, growth = _csx.C.c_int32(-1), _csx.C.c_int32(-1), _csx.C.c_double(-1.0)
    _csx.check(lib.csx_cholsol_info(plan, path, None, None))
    assert path.value == 4
    _csx.check(lib.csx_cholsol_sn_info(plan, None, None, None, mc, growth))
    assert mc.value == cores
    assert (growth.value <= 1e3) == bool(cores) and growth.value > 10
    for k in (1, 5):
        B = synth.rhs(n, k, 4)
        X = cs.dvec(B)
        _csx.check(lib.csx_cholsol_solve(plan, X.handle, k))
        Xn = X.numpy().reshape(n, k)
        for r in range(k):
            ref = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r].copy()))
            assert np.max(np.abs(Xn[:, r] - ref)) <= 1e-10 * np.max(np.abs(ref))
    _csx.check(lib.csx_free(plan))
    _csx.check(lib.csx_free(hL))


def test_supernodal_schedule_refuses_a_triangle_that_is_not_a_cholesky_factor(cs):
    """csx_cholsol_plan takes any lower triangle with the diagonal first.  One whose columns look like supernodes by
    their counts but do not share their rows (or whose rows are not ancestors in its own tree) must not get the
    supernodal schedule: k_sn_verify sends it back to the level-scheduled plans, and the answer is still right."""
    import _csx
    lib = _csx.lib()
    n = 400
    rng = np.random.default_rng(3)
    cols_i, cols_x, Ap = [], [], [0]
    for j in range(n):
        below = np.arange(j + 1, n)
        # j + 1 is always the first off-diagonal row (a chain tree) and the counts fall by one from column to column
        # within runs of eight, but the other rows are random: not the rows of the previous column
        cnt = min(len(below), 12 - (j % 8))
        pick = below[:1].tolist() + sorted(rng.choice(below[1:], size=max(0, min(cnt - 1, len(below) - 1)), replace=False).tolist()) if len(below) else []
        rows = [j] + pick
        cols_i.append(np.asarray(rows, np.int32))
        cols_x.append(np.concatenate([[4.0 + len(rows)], rng.uniform(-1, 1, len(rows) - 1)]))
        Ap.append(Ap[-1] + len(rows))
    Lp, Li, Lx = np.asarray(Ap, np.int32), np.concatenate(cols_i), np.concatenate(cols_x)
    hL = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Lp), _csx.pi(Li), _csx.pd(Lx), hL))
    b = synth.rhs(n, 1, 0)[:, 0]
    ref = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b))
    # "tri.supernodes" = 2: fundamental supernodes only -- the columns' counts say "supernode", their rows do not: refused.
    # Default: the tree is a chain, every later row IS an ancestor, so runs of the chain as RELAXED supernodes (which ask
    # nothing of the rows) are a valid schedule -- accepted, and the answer has to be right either way.
    for mode, want_path in ((2, 0), (1, 4)):
        with _csx.option("tri.supernodes", mode):
            plan = _csx.new_handle()
            _csx.check(lib.csx_cholsol_plan(hL, None, plan))
            _csx.check(lib.csx_cholsol_set_order(plan, 0))
            path = _csx.C.c_int32(-1)
            _csx.check(lib.csx_cholsol_info(plan, path, None, None))
            assert path.value == want_path, mode
            X = cs.dvec(b)
            _csx.check(lib.csx_cholsol_solve(plan, X.handle, 1))
            assert np.max(np.abs(X.numpy() - ref)) <= 1e-10 * np.max(np.abs(ref)), mode
            _csx.free(plan)
    _csx.free(hL)


@pytest.mark.parametrize("bs", [8, 16, 32, 64])
def test_exact_dense_block_kernel_variants_all_have_the_reference_bits(cs, bs):
    """The default (exact) order on forests of dense blocks has six kernel variants (the L values by DPP row broadcast, with or
    without one term in four by an LDS broadcast read -- the default is with -- or by LDS broadcast with one fence per row or through a register ring, one or two right-hand sides
    per lane; "cholsol.exact_variant" forces one).  Each must be bit-identical to cs_lsolve + cs_ltsolve of the oracle, for 130
    right-hand sides (two full groups and a partial one; a partial pair for the two-per-lane variants)."""
    import _csx
    nblocks, k = 9, 130
    n = nblocks * bs
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 11)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, exact=True)
    assert F.info()["dense_block"] == bs
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    B = synth.rhs(n, k, 3)
    ref = {r: CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r])) for r in (0, 63, 64, 127, 128, 129)}
    for variant in (0, 1, 2, 3, 4, 5, 6):
        with _csx.option("cholsol.exact_variant", variant):
            dB = cs.dvec(B)
            assert F.solve(dB)
            X = dB.numpy()
        for r, v in ref.items():
            assert X[:, r].tobytes() == v.tobytes(), (variant, r)
    # the default (5: L values by DPP row broadcast) lets the waves that solve one block share its LDS copy: four, two or
    # one block(s) per workgroup as the number of 64-wide groups of right-hand sides allows (130 -> 3 groups: four blocks,
    # 128 -> 2: two, 256 -> 4: one); 9 blocks leave the last workgroup partly empty in every case.  A right-hand side
    # with zeros and negative zeros in it: the products' signs of zero must come out as the reference's.
    for kk in (128, 256, 1):
        Bk = synth.rhs(n, kk, 5)
        Bk[::3, 0] = 0.0
        Bk[1::7, 0] = -0.0
        dB = cs.dvec(Bk)
        assert F.solve(dB)
        X = dB.numpy().reshape(n, kk)
        for r in sorted({0, kk // 2, kk - 1}):
            v = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, Bk[:, r].copy()))
            assert X[:, r].tobytes() == v.tobytes(), (kk, r)
