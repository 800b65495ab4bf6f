"""Small dependency components on the matrix cores (csx_trimfma.hip, round 5): the ROUNDING-EQUAL order of the triangular
solves (cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve, csparse.py:1330-1365, :2368-2385, :2460-2475) on factors that fall into
many small independent components, and of cs_cholsol's solve phase (:640-643) on forests of small trees that are not equal dense
blocks.  Every column against the plain-C oracle at BASELINE's 1e-10 (componentwise, SURVEY 8d's measure with the terms of the
substitution: tests/tol.py); the exact order on the same plans stays bit-identical."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as CO
import synth
import tol as TOL
from test_gpu_cholclique import _blocks, _tree_blocks
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs, cs  # noqa: F401
from test_gpu_tricomponents import _block_tri

pytestmark = pytest.mark.gpu

KINDS = {"lsolve": 0, "ltsolve": 1, "usolve": 2, "utsolve": 3}


def _terms(n, Tp, Ti, Tx, x, b, kind):
    """sum of |terms| behind every unknown of T x = b (or T' x = b): |b_i| + sum_j |T_ij| |x_j|, over |T_ii|"""
    T = TOL.csc(n, Tp, Ti, Tx)
    if kind in ("ltsolve", "utsolve"):
        T = T.T
    d = np.abs(T.diagonal())
    S = abs(T)
    return (np.abs(b) + S @ np.abs(x) - d * np.abs(x)) / d


def _tri_order(plan):
    import _csx
    mc, g = C.c_int32(-1), C.c_double(-1.0)
    _csx.check(_csx.lib().csx_tri_order_info(plan, mc, g))
    return mc.value, g.value


@pytest.mark.parametrize("nrhs", [9, 64, 70, 130])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_components_on_the_matrix_cores(cs, kind, nrhs):
    """Blocks of 67 / 1 / 5 / 80 / 64 / 2 / 17 / 33 / 48 rows with random patterns (every size class, padding in each): the
    rounding-equal order against the oracle; then the SAME plan back in the exact order, bit for bit."""
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(KINDS[kind] * 100 + nrhs)
    lower = kind in ("lsolve", "ltsolve")
    n, Tp, Ti, Tx = _block_tri(rng, 300, [67, 1, 5, 80, 64, 2, 17, 33, 48], 0.2, lower, True)
    T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    B = synth.rhs(n, nrhs, 0)
    refs = np.stack([getattr(CO, kind)(n, Tp, Ti, Tx, B[:, r]) for r in range(nrhs)], axis=1)
    X = cs.dvec(B)
    assert getattr(cs, "cs_" + kind)(T, X) is True                   # makes the plan (exact order)
    assert X.numpy().reshape(n, -1).tobytes() == refs.tobytes()
    plan = T._dev.plans[KINDS[kind]]
    assert _tri_order(plan)[0] == 0
    _csx.check(lib.csx_tri_set_order(plan, 0))
    Y = cs.dvec(B)
    _csx.check(lib.csx_tri_solve(plan, Y.handle, nrhs))
    mc, growth = _tri_order(plan)
    assert mc == 1 and 1.0 <= growth <= 1e3
    got = Y.numpy().reshape(n, -1)
    assert not np.array_equal(got, refs)                              # it really is another order of operations
    for r in (0, nrhs // 2, nrhs - 1):
        assert TOL.componentwise(got[:, r], refs[:, r], _terms(n, Tp, Ti, Tx, refs[:, r], B[:, r], kind)) <= TOL.X_RTOL
    assert TOL.normwise(got, refs) <= 1e-13
    # few right-hand sides stay with the exact kernels even in this order (a matrix-core tile is 16 wide)
    Z = cs.dvec(B[:, :3].copy())
    _csx.check(lib.csx_tri_solve(plan, Z.handle, 3))
    assert Z.numpy().reshape(n, 3).tobytes() == refs[:, :3].copy().tobytes()
    _csx.check(lib.csx_tri_set_order(plan, 1))
    X2 = cs.dvec(B)
    _csx.check(lib.csx_tri_solve(plan, X2.handle, nrhs))
    assert X2.numpy().reshape(n, -1).tobytes() == refs.tobytes()


def test_components_wider_than_80_rows_or_ill_conditioned_keep_the_exact_kernels(cs):
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(5)
    n, Tp, Ti, Tx = _block_tri(rng, 200, [67, 130, 9], 0.15, True, True)       # a block of 130 rows
    T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    B = synth.rhs(n, 40, 0)
    refs = np.stack([CO.lsolve(n, Tp, Ti, Tx, B[:, r]) for r in range(40)], axis=1)
    X = cs.dvec(B)
    assert cs.cs_lsolve(T, X)
    plan = T._dev.plans[cs.TRI_L]
    _csx.check(lib.csx_tri_set_order(plan, 0))
    Y = cs.dvec(B)
    _csx.check(lib.csx_tri_solve(plan, Y.handle, 40))
    assert _tri_order(plan)[0] == 0 and Y.numpy().reshape(n, -1).tobytes() == refs.tobytes()
    # a diagonal tile whose inverse is large: bidiagonal blocks with -2 below a unit diagonal (inverse entries 2^k)
    nb, m = 100, 48
    cols_i, cols_x, Ap = [], [], [0]
    for b in range(nb):
        for c in range(m):
            rows = [c] + ([c + 1] if c + 1 < m else [])
            vals = [1.0] + ([-2.0] if c + 1 < m else [])
            cols_i.append(np.asarray(rows, np.int32) + b * m)
            cols_x.append(np.asarray(vals))
            Ap.append(Ap[-1] + len(rows))
    n2 = nb * m
    Tp2, Ti2, Tx2 = np.asarray(Ap, np.int32), np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)
    T2 = cs.cs_pin(_host_cs(cs, n2, n2, Tp2, Ti2, Tx2))
    B2 = synth.rhs(n2, 20, 0)
    ref2 = np.stack([CO.lsolve(n2, Tp2, Ti2, Tx2, B2[:, r]) for r in range(20)], axis=1)
    X2 = cs.dvec(B2)
    assert cs.cs_lsolve(T2, X2)
    plan2 = T2._dev.plans[cs.TRI_L]
    _csx.check(lib.csx_tri_set_order(plan2, 0))
    Y2 = cs.dvec(B2)
    _csx.check(lib.csx_tri_solve(plan2, Y2.handle, 20))
    mc, growth = _tri_order(plan2)
    assert mc == 0 and growth > 1e3                       # refused by the guard
    assert Y2.numpy().reshape(n2, -1).tobytes() == ref2.tobytes()


def test_duplicate_rows_and_explicit_zeros_in_a_column(cs):
    """cs_lu's factors may name a row twice in a column and hold explicit zeros: both products are subtracted, so the
    dense form carries the SUM of the coefficients."""
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(8)
    n, Tp, Ti, Tx = _block_tri(rng, 150, [20, 37], 0.3, True, True)
    cols = [(Ti[Tp[c]:Tp[c + 1]].tolist(), Tx[Tp[c]:Tp[c + 1]].tolist()) for c in range(n)]
    dup = 0
    for c in range(n):
        if len(cols[c][0]) >= 3 and c % 3 == 0:
            cols[c][0].append(cols[c][0][1]), cols[c][1].append(0.25)         # the first off-diagonal row again
            cols[c][0].append(cols[c][0][2]), cols[c][1].append(0.0)          # an explicit zero on another
            dup += 1
    assert dup > 100
    Tp2 = np.zeros(n + 1, np.int32)
    Tp2[1:] = np.cumsum([len(c[0]) for c in cols])
    Ti2 = np.concatenate([c[0] for c in cols]).astype(np.int32)
    Tx2 = np.concatenate([c[1] for c in cols])
    T = cs.cs_pin(_host_cs(cs, n, n, Tp2, Ti2, Tx2))
    B = synth.rhs(n, 33, 1)
    for kind in ("lsolve", "ltsolve"):
        refs = np.stack([getattr(CO, kind)(n, Tp2, Ti2, Tx2, B[:, r]) for r in range(33)], axis=1)
        X = cs.dvec(B)
        assert getattr(cs, "cs_" + kind)(T, X) and X.numpy().reshape(n, -1).tobytes() == refs.tobytes()
        plan = T._dev.plans[KINDS[kind]]
        _csx.check(lib.csx_tri_set_order(plan, 0))
        Y = cs.dvec(B)
        _csx.check(lib.csx_tri_solve(plan, Y.handle, 33))
        assert _tri_order(plan)[0] == 1
        assert TOL.normwise(Y.numpy().reshape(n, -1), refs) <= 1e-13
        _csx.check(lib.csx_tri_set_order(plan, 1))


def test_lusol_factor_on_W_blocks_rounding_equal_lists_exact(cs):
    """BASELINE config 3 at a fifth of its size: lusol_factor's default solves a dvec block in the rounding-equal order (both
    triangular solves on the matrix cores), a list -- and everything with exact=True -- with the bits of cs_lusol."""
    n, Ap, Ai, Ax = _w_matrix(300)
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    F = cs.lusol_factor(A, 0, 1.0)
    FL, FU = F.factors.L, F.factors.U
    Lp, Li, Lx = (np.asarray(v) for v in (FL.p, FL.i, FL.x))
    Up, Ui, Ux = (np.asarray(v) for v in (FU.p, FU.i, FU.x))
    Lp, Li, Up, Ui = (v.astype(np.int32) for v in (Lp, Li, Up, Ui))
    pinv = np.asarray(F.factors.pinv)
    b = 1.0 + np.arange(n) / n
    k = 70
    scales = 1.0 + 0.5 * np.arange(k)
    B = np.ascontiguousarray(b[:, None] * scales[None, :])

    def oracle(col):
        pb = np.empty(n)
        pb[pinv] = col
        y = CO.lsolve(n, Lp, Li, Lx, pb)
        return y, CO.usolve(n, Up, Ui, Ux, y)

    dB = cs.dvec(B)
    assert F.solve(dB) is True
    info = F.info()
    assert info["L"]["matrix_cores"] and info["U"]["matrix_cores"]
    X = dB.numpy().reshape(n, k)
    for r in (0, 33, k - 1):
        y, want = oracle(B[:, r])
        terms = _terms(n, Up, Ui, Ux, want, y, "usolve")
        assert TOL.componentwise(X[:, r], want, terms) <= TOL.X_RTOL, r
        assert TOL.normwise(X[:, r], want) <= 1e-12
    one = B[:, 33].tolist()
    assert F.solve(one) is True
    assert np.asarray(one).tobytes() == oracle(B[:, 33])[1].tobytes()
    Fe = cs.lusol_factor(A, 0, 1.0, exact=True)
    dBe = cs.dvec(B)
    assert Fe.solve(dBe) is True
    Xe = dBe.numpy().reshape(n, k)
    for r in (0, 33, k - 1):
        assert Xe[:, r].tobytes() == oracle(B[:, r])[1].tobytes()
    blk = cs.dvec(B)                                       # the reference's driver on a block: always exact
    assert cs.cs_lusol(0, A, blk, 1.0) is True and blk.numpy().tobytes() == dBe.numpy().tobytes()


@pytest.mark.parametrize("shape", ["unequal_cliques", "arrow", "random_trees", "tridiagonal"])
def test_cholsol_forests_of_small_trees_on_the_matrix_cores(cs, shape):
    """cs_cholsol's solve phase on forests that are NOT equal dense blocks: cliques of 1 .. 64 columns, small sparse trees.
    Rounding-equal order: csx_cholsol_info path 5, every column within 1e-10 of cs_lsolve + cs_ltsolve on the same L (the
    oracle's loops); the exact order on the same solver bit-identical."""
    import _csx
    rng = np.random.default_rng(3)
    sizes = list(rng.integers(1, 65, 260)) + [64, 1, 2, 16, 17, 48, 49]
    if shape == "unequal_cliques":
        n, Ap, Ai, Ax = _blocks(sizes, 21)
    else:
        n, Ap, Ai, Ax = _tree_blocks(sizes, 22, {"arrow": "arrow", "random_trees": "random", "tridiagonal": "tridiagonal"}[shape])
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    parent, cp = CO.schol(n, Ap, Ai)
    k = 70
    B = synth.rhs(n, k, 2)
    F = cs.cholsol_factor(A)                              # default: blocks rounding-equal, lists exact
    gLp, gLi, gLx = (np.asarray(v) for v in (F.L.p, F.L.i, F.L.x))
    gLp, gLi = gLp.astype(np.int32), gLi.astype(np.int32)
    gLx = gLx[:gLp[n]]
    gLi = gLi[:gLp[n]]
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    info = F.info()
    assert info["matrix_cores"] and info["dense_block"] == 0 and info["fused_local"]
    path = C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_info(F.plan_handle, path, None, None))
    assert path.value == 5
    X = dB.numpy().reshape(n, k)
    for r in (0, 35, k - 1):
        y = CO.lsolve(n, gLp, gLi, gLx, B[:, r])
        ref = CO.ltsolve(n, gLp, gLi, gLx, y)
        assert TOL.componentwise(X[:, r], ref, TOL.cholsolve_terms(n, gLp, gLi, gLx, y, ref)) <= TOL.X_RTOL, (shape, r)
        assert TOL.normwise(X[:, r], ref) <= 1e-13
    one = B[:, 35].tolist()
    assert F.solve(one) is True                            # a list: the reference's order
    ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, 35]))
    assert np.asarray(one).tobytes() == ref.tobytes()
    with _csx.option("cholsol.dense_blocks", 0):           # the fused per-tree kernel on the same plan, rounding-equal order asked
        dB1 = cs.dvec(B)
        assert F.solve(dB1) is True
        assert dB1.numpy().reshape(n, k)[:, 35].tobytes() == ref.tobytes()


@pytest.mark.parametrize("k", [1, 70, 128, 200])
def test_exact_order_on_unequal_cliques_by_padded_size_classes(cs, k):
    """The exact order on a forest of cliques of 1 .. 64 columns (csx_cholsol_factor's plan): blocks bucketed by size class
    8 / 16 / 32 / 64 and padded at their end with the identity, the register-resident exact kernel per class.  The padding changes
    no bit: every column equals cs_lsolve + cs_ltsolve on the same L (the oracle's loops), for 1, 70, 128 and 200 right-hand sides
    (every sharing mode of the kernel), zeros and negative zeros in the right-hand side included."""
    import _csx
    rng = np.random.default_rng(11)
    sizes = list(rng.integers(1, 65, 200)) + [64, 1, 8, 9, 16, 17, 32, 33]
    n, Ap, Ai, Ax = _blocks(sizes, 23)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, exact=True)
    path = C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_info(F.plan_handle, path, None, None))
    gLp, gLi, gLx = (np.asarray(v) for v in (F.L.p, F.L.i, F.L.x))
    gLp, gLi = gLp.astype(np.int32), gLi.astype(np.int32)
    gLi, gLx = gLi[:gLp[n]], gLx[:gLp[n]]
    B = synth.rhs(n, k, 2) if k > 1 else synth.rhs(n, 1, 2)
    B = B.reshape(n, -1).copy()
    B[::7, 0] = 0.0
    B[3::11, -1] = -0.0
    dB = cs.dvec(B if k > 1 else B[:, 0].copy())
    assert F.solve(dB) is True
    X = dB.numpy().reshape(n, -1)
    for r in sorted({0, k // 2, k - 1}):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert X[:, r].tobytes() == ref.tobytes(), r
    with _csx.option("cholsol.dense_blocks", 0):            # the fused per-tree kernel (the general plan) on the same solver
        dB1 = cs.dvec(B if k > 1 else B[:, 0].copy())
        assert F.solve(dB1) is True
        assert dB1.numpy().tobytes() == dB.numpy().tobytes()


@pytest.mark.parametrize("nrhs", [9, 70, 128, 192])
@pytest.mark.parametrize("perms", ["both", "pinv_only", "q_only", "none"])
def test_lusol_solve_fuses_the_permutations_into_the_sweeps(cs, nrhs, perms):
    """csx_lusol_solve (cs_lusol's solve phase, csparse.py:1470-1473) on block-triangular L and U with permutations that move rows
    across the whole block: in the rounding-equal order on forests of small components the sweep over L gathers through pinv and
    the sweep over U scatters through q (whole chunks of 64 right-hand sides: the buffer-resource gather / scatter; other counts: row
    look-ups) -- every column BIT-identical to the four separate calls it replaces (the arithmetic is the same, only the data
    movement differs), and those inside BASELINE's 1e-10 of the oracle; in the exact order the same fusion runs through the in-LDS
    sweeps for more than 32 right-hand sides (the four steps below that), bits of the reference either way."""
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(nrhs * 7 + len(perms))
    sizes = [67, 1, 5, 80, 64, 2, 17, 33, 48]

    def linked(lower):
        # every column linked to its neighbour inside the block: a block is ONE component on consecutive rows
        cols_i, cols_x, Ap, base = [], [], [0], 0
        for b in range(260):
            m = sizes[b % len(sizes)]
            for c in range(m):
                cand = np.arange(c + 2, m) if lower else np.arange(0, c - 1)
                pick = cand[rng.random(len(cand)) < 0.2]
                rng.shuffle(pick)
                link = [c + 1] if lower and c + 1 < m else ([c - 1] if not lower and c >= 1 else [])
                d = rng.uniform(1.0, 2.0) * (1 if rng.random() < 0.5 else -1)
                off_r = link + pick.tolist()
                off_v = rng.uniform(-0.5, 0.5, len(link)).tolist() + rng.uniform(-0.1, 0.1, len(pick)).tolist()
                rows = ([c] + off_r) if lower else (off_r + [c])
                vals = ([d] + off_v) if lower else (off_v + [d])
                cols_i.append(np.asarray(rows, np.int32) + base)
                cols_x.append(np.asarray(vals))
                Ap.append(Ap[-1] + len(rows))
            base += m
        return base, np.asarray(Ap, np.int32), np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)
    n, Lp, Li, Lx = linked(True)
    n2, Up, Ui, Ux = linked(False)
    assert n2 == n
    L = cs.cs_pin(_host_cs(cs, n, n, Lp, Li, Lx))
    U = cs.cs_pin(_host_cs(cs, n, n, Up, Ui, Ux))
    pinv = rng.permutation(n).astype(np.int32) if perms in ("both", "pinv_only") else None
    q = rng.permutation(n).astype(np.int32) if perms in ("both", "q_only") else None
    B = synth.rhs(n, nrhs, 3)

    def oracle(col):
        pb = np.empty(n)
        pb[pinv if pinv is not None else np.arange(n)] = col
        y = CO.lsolve(n, Lp, Li, Lx, pb)
        x = CO.usolve(n, Up, Ui, Ux, y)
        out = np.empty(n)
        out[q if q is not None else np.arange(n)] = x
        return y, x, out

    X = cs.dvec(B[:, :1].copy())
    assert cs.cs_lsolve(L, X) is True and cs.cs_usolve(U, X) is True        # makes the two plans
    pl, pu = L._dev.plans[KINDS["lsolve"]], U._dev.plans[KINDS["usolve"]]

    def ivec(p):
        if p is None:
            return _csx.H(0)
        h = _csx.new_handle()
        _csx.check(lib.csx_ivec_upload(_csx.pi(p), n, h))
        return h
    hp, hq = ivec(pinv), ivec(q)
    try:
        results = {}
        for exact in (1, 0):
            for plan in (pl, pu):
                _csx.check(lib.csx_tri_set_order(plan, exact))
            b, w = cs.dvec(B), cs.dvec(n, nrhs)
            fused = C.c_int(-1)
            _csx.check(lib.csx_lusol_solve(pl, pu, hp, hq, b.handle, w.handle, nrhs, C.byref(fused)))
            assert fused.value == (1 if (exact == 0 and nrhs > 8) or (exact == 1 and nrhs > 32) else 0)   # (exact: the in-LDS sweeps of many right-hand sides take the row maps too)
            results[exact] = b.numpy().reshape(n, nrhs).copy()
            # the four calls it replaces, same orders
            b2, x2 = cs.dvec(B), cs.dvec(n, nrhs)
            _csx.check(lib.csx_permute_vec(hp, b2.handle, x2.handle, n, nrhs, 1))
            _csx.check(lib.csx_tri_solve(pl, x2.handle, nrhs))
            _csx.check(lib.csx_tri_solve(pu, x2.handle, nrhs))
            _csx.check(lib.csx_permute_vec(hq, x2.handle, b2.handle, n, nrhs, 1))
            assert b2.numpy().tobytes() == results[exact].tobytes(), (exact, "fused and separate calls differ")
        for r in sorted({0, nrhs // 2, nrhs - 1}):
            y, x, want = oracle(B[:, r])
            assert results[1][:, r].tobytes() == want.tobytes()
            terms = np.empty(n)
            terms[q if q is not None else np.arange(n)] = _terms(n, Up, Ui, Ux, x, y, "usolve")
            assert TOL.componentwise(results[0][:, r], want, terms) <= TOL.X_RTOL
    finally:
        for plan in (pl, pu):
            lib.csx_tri_set_order(plan, 1)
        for h in (hp, hq):
            if h.value:
                _csx.free(h)


@pytest.mark.parametrize("order", [0, 1])
def test_whole_chunks_and_partial_chunks_of_right_hand_sides_give_the_same_bits(cs, order):
    """k_rag_mfma moves X in one of five ways (MODE: run-time paths; consecutive rows through a buffer resource; gather; scatter;
    looked-up rows through whole-block resources) chosen by the host from the number of right-hand sides, the rows of the components
    and the permutation -- the arithmetic of a right-hand side is the same in all of them: a batch of 128 (whole chunks: MODE 1 in
    the natural order, MODE 4 under a fill-reducing permutation) and a batch of 70 (MODE 0) agree bit for bit on the columns they
    share, and both are inside the budget against the oracle."""
    rng = np.random.default_rng(11)
    sizes = list(rng.integers(1, 65, 200)) + [64, 33, 48, 17]
    n, Ap, Ai, Ax = _blocks(sizes, 23)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, order=order, exact=False)
    assert F.info()["matrix_cores"]
    B = synth.rhs(n, 128, 5)
    d128, d70 = cs.dvec(B), cs.dvec(np.ascontiguousarray(B[:, :70]))
    assert F.solve(d128) is True and F.solve(d70) is True
    X128, X70 = d128.numpy().reshape(n, 128), d70.numpy().reshape(n, 70)
    assert X128[:, :70].tobytes() == X70.tobytes()
    Fe = cs.cholsol_factor(A, order=order, exact=True)
    de = cs.dvec(np.ascontiguousarray(B[:, :3]))
    assert Fe.solve(de) is True
    Xe = de.numpy().reshape(n, 3)
    for r in range(3):
        assert TOL.normwise(X128[:, r], Xe[:, r]) <= 1e-12
        R = CO.gaxpy(n, n, Ap, Ai, Ax, X128[:, r], -B[:, r])
        assert np.max(np.abs(R)) <= 1e-11 * np.max(np.abs(B[:, r])) * 64
