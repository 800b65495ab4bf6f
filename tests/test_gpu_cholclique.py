"""Forests of cliques on consecutive columns (csx_cholclique.hip): cs_schol / cs_chol / the cholsol plan recognise
block-diagonal matrices with dense blocks from the matrix itself and skip the general pattern machine.  Everything is
compared with the plain-C oracle (cs_schol :2051-2072, cs_chol :561-619) and with the general path ("chol.clique" = 0)."""
import numpy as np
import pytest

import c_oracle as CO
import synth
from test_gpu_parity import _host_cs, cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _arr(A):
    nnz = A.p[A.n]
    return (np.asarray(A.p, np.int32), np.asarray(A.i[:nnz], np.int32), np.asarray(A.x[:nnz], np.float64))


def _blocks(sizes, seed, density=1.0, shuffle_lower=False, extra=None):
    """Block-diagonal SPD matrix, full symmetric storage.  Block b: R R' / bs + bs I with entries dropped to `density`
    (never from the block's first row / column, so the factor's block stays a clique).  Columns ascending unless
    shuffle_lower (then the rows BELOW the diagonal come in a random order: cs_chol never reads them)."""
    rng = np.random.default_rng(seed)
    n = int(sum(sizes))
    cols_i, cols_x = [], []
    a = 0
    for bs in sizes:
        R = rng.uniform(-1.0, 1.0, (bs, bs))
        B = R @ R.T / bs + bs * np.eye(bs)
        keep = rng.uniform(size=(bs, bs)) < density
        keep = np.triu(keep) | np.triu(keep).T
        keep[0, :] = keep[:, 0] = True
        keep[np.arange(bs), np.arange(bs)] = True
        B = B * keep
        for c in range(bs):
            rows = np.nonzero(keep[:, c])[0]
            if shuffle_lower:
                lo = rows[rows > c]
                rng.shuffle(lo)
                rows = np.concatenate([rows[rows <= c], lo])
            cols_i.append(rows + a)
            cols_x.append(B[rows, c])
        a += bs
    if extra:
        extra(cols_i, cols_x)
    Ap = np.zeros(n + 1, np.int32)
    Ap[1:] = np.cumsum([len(r) for r in cols_i])
    return n, Ap, np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)


def _factor_both_ways(cs, n, Ap, Ai, Ax):
    import _csx
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_schol(0, A)
    N = cs.cs_chol(A, S)
    with _csx.option("chol.clique", 0):
        S0 = cs.cs_schol(0, A)
        N0 = cs.cs_chol(A, S0)
    return A, S, N, S0, N0


@pytest.mark.parametrize("case", ["mixed", "sparse_blocks", "ones", "one_block_64", "lower_shuffled", "lower_in_front"])
def test_clique_forest_factor_has_the_oracle_bits(cs, case):
    """"mixed", "ones", "one_block_64": dense blocks with sorted columns -- the block kernel does not even read the row indices
    (k_clique_min found every upper part dense and stored in front); "lower_in_front": the same blocks with a lower entry
    moved to the front of some columns -- that shortcut must be refused; "sparse_blocks", "lower_shuffled": fill inside the
    blocks, lower parts in any order."""
    rng = np.random.default_rng(7)
    if case == "mixed":
        sizes, dens, shuf = list(rng.integers(1, 65, 300)) + [64, 63, 1, 2, 17], 1.0, False
    elif case == "sparse_blocks":
        sizes, dens, shuf = list(rng.integers(2, 65, 200)), 0.3, False
    elif case == "ones":
        sizes, dens, shuf = [1] * 500, 1.0, False
    elif case == "one_block_64":
        sizes, dens, shuf = [64], 1.0, False
    elif case == "lower_in_front":
        sizes, dens, shuf = list(rng.integers(2, 65, 120)), 1.0, False
    else:
        sizes, dens, shuf = list(rng.integers(1, 50, 150)), 0.6, True

    def lower_first(cols_i, cols_x):
        moved = 0
        for c in range(len(cols_i)):
            if c % 5 == 0 and len(cols_i[c]) and cols_i[c][-1] > c:        # the column's last entry is a lower one: to the front
                cols_i[c] = np.concatenate([cols_i[c][-1:], cols_i[c][:-1]])
                cols_x[c] = np.concatenate([cols_x[c][-1:], cols_x[c][:-1]])
                moved += 1
        assert moved > 10

    n, Ap, Ai, Ax = _blocks(sizes, 11, dens, shuf, lower_first if case == "lower_in_front" else None)
    A, S, N, S0, N0 = _factor_both_ways(cs, n, Ap, Ai, Ax)
    parent, cp = CO.schol(n, Ap, Ai)
    assert S.parent == parent.tolist() == S0.parent and S.cp == cp.tolist() == S0.cp and S.lnz == int(cp[n])
    # the forest really is one chain per block with full columns
    assert int(cp[n]) == sum(b * (b + 1) // 2 for b in sizes)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    gp, gi, gx = _arr(N.L)
    assert gp.tolist() == Lp.tolist() and gi.tolist() == Li.tolist()
    assert gx.tobytes() == Lx.tobytes()                      # block kernel: the reference's operation order
    g0p, g0i, g0x = _arr(N0.L)
    assert g0p.tolist() == Lp.tolist() and g0i.tolist() == Li.tolist()
    assert np.max(np.abs(g0x - Lx)) <= 1e-13 * np.abs(Lx).max()


@pytest.mark.parametrize("case", ["duplicates", "upper_unsorted", "block_of_100", "reaches_back", "not_full"])
def test_what_is_not_a_clique_forest_takes_the_general_path(cs, case):
    """Same answers either way: the recognition must refuse (or hand over) and the general machine runs."""
    rng = np.random.default_rng(3)
    sizes = list(rng.integers(2, 40, 60))

    def extra(cols_i, cols_x):
        if case == "duplicates":            # A(0, 5) stored twice in column 5 of the first big block
            b = int(np.argmax(np.asarray(sizes) >= 8))
            c = int(sum(sizes[:b])) + 5
            cols_i[c] = np.concatenate([cols_i[c][:1], cols_i[c][:1], cols_i[c][1:]])
            cols_x[c] = np.concatenate([cols_x[c][:1] * 0.25, cols_x[c][:1] * 0.75, cols_x[c][1:]])
        elif case == "upper_unsorted":
            for c in range(len(cols_i)):
                order = rng.permutation(len(cols_i[c]))
                cols_i[c], cols_x[c] = cols_i[c][order], cols_x[c][order]
        elif case == "reaches_back":        # one entry links the third block to the first: trees merge
            a2 = sizes[0] + sizes[1]
            cols_i[a2] = np.concatenate([[0], cols_i[a2]])
            cols_x[a2] = np.concatenate([[1e-3], cols_x[a2]])
            cols_i[0] = np.concatenate([cols_i[0], [a2]])
            cols_x[0] = np.concatenate([cols_x[0], [1e-3]])
        elif case == "not_full":            # A(first, last) of a block dropped: its first column of L is not full
            b = int(np.argmax(np.asarray(sizes) >= 4))
            a, bs = sum(sizes[:b]), sizes[b]
            last = a + bs - 1
            keep = cols_i[last] != a
            cols_i[last], cols_x[last] = cols_i[last][keep], cols_x[last][keep]
            keep = cols_i[a] != last
            cols_i[a], cols_x[a] = cols_i[a][keep], cols_x[a][keep]

    if case == "block_of_100":
        sizes = [100, 3, 70]
    n, Ap, Ai, Ax = _blocks(sizes, 5, 1.0, False, extra if case != "block_of_100" else None)
    A, S, N, S0, N0 = _factor_both_ways(cs, n, Ap, Ai, Ax)
    parent, cp = CO.schol(n, Ap, Ai)
    assert S.parent == parent.tolist() == S0.parent and S.cp == cp.tolist() == S0.cp
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    for M in (N, N0):
        gp, gi, gx = _arr(M.L)
        assert gp.tolist() == Lp.tolist() and gi.tolist() == Li.tolist()
        assert np.max(np.abs(gx - Lx)) <= 1e-13 * np.abs(Lx).max()
    if case != "not_full":       # (a block that is a tree but no clique: the forest kernel, rounding-equal to the general one)
        assert _arr(N.L)[2].tobytes() == _arr(N0.L)[2].tobytes()


def _tree_blocks(sizes, seed, kind, missing_diagonal=None):
    """Block-diagonal SPD matrix whose blocks are SPARSE (full symmetric storage, columns ascending): "tridiagonal" (a chain
    with no fill), "arrow" (tridiagonal + a dense last row: chain, still no fill), "random" (a few entries per row: branching
    trees with fill), "star" (every row linked to the block's LAST row only: a star, no fill)."""
    rng = np.random.default_rng(seed)
    n = int(sum(sizes))
    cols_i, cols_x = [], []
    a = 0
    for bs in sizes:
        K = np.eye(bs, dtype=bool)
        idx = np.arange(bs - 1)
        if kind in ("tridiagonal", "arrow"):
            K[idx, idx + 1] = True
        if kind in ("arrow", "star"):
            K[:, bs - 1] = True
        if kind == "random":
            K |= np.triu(rng.uniform(size=(bs, bs)) < 2.5 / bs)
        K = np.triu(K) | np.triu(K).T
        V = rng.uniform(-1.0, 1.0, (bs, bs))
        B = (V + V.T) * K
        B[np.arange(bs), np.arange(bs)] = np.abs(B).sum(axis=1) + 1.0          # strictly diagonally dominant
        for c in range(bs):
            rows = np.nonzero(K[:, c])[0]
            if missing_diagonal is not None and a + c == missing_diagonal:
                rows = rows[rows != c]
            cols_i.append(rows + a)
            cols_x.append(B[rows, c])
        a += bs
    Ap = np.zeros(n + 1, np.int32)
    Ap[1:] = np.cumsum([len(r) for r in cols_i])
    return n, Ap, np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)


def _chol_path(cs):
    import ctypes as C
    import _csx
    path = C.c_int32(-1)
    _csx.check(_csx.lib().csx_chol_info(path, None))
    return path.value


@pytest.mark.parametrize("kind", ["tridiagonal", "arrow", "random", "star"])
def test_forest_of_small_sparse_trees_is_analysed_and_factored_by_one_wave_per_block(cs, kind):
    """Blocks of consecutive columns closed under their upper entries, at most 64 columns each: elimination tree and column
    counts come from the symbolic elimination on row masks (k_forest_symbolic) and must be the reference's (cs_schol
    :2051-2072); the block kernel factors them as dense triangles and stores the pattern's rows.  Chains ("tridiagonal",
    "arrow") are eliminated in the reference's order: the reference's bits.  Branching trees agree to rounding, like the
    general column kernels ("chol.forest" = 0) -- the reference sums a row's updates in cs_ereach's order, which no
    right-looking kernel follows."""
    import _csx
    rng = np.random.default_rng(21)
    sizes = list(rng.integers(1, 65, 250)) + [64, 1, 2, 64, 33]
    n, Ap, Ai, Ax = _tree_blocks(sizes, 5, kind)
    parent, cp = CO.schol(n, Ap, Ai)
    assert int(cp[n]) < sum(b * (b + 1) // 2 for b in sizes)          # really sparse inside the blocks
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_schol(0, A)
    assert S.parent == parent.tolist() and S.cp == cp.tolist() and S.lnz == int(cp[n])
    N = cs.cs_chol(A, S)
    assert _chol_path(cs) == 2
    gp, gi, gx = _arr(N.L)
    assert gp.tolist() == Lp.tolist() and gi.tolist() == Li.tolist()
    if kind in ("tridiagonal", "arrow"):
        assert gx.tobytes() == Lx.tobytes()
    assert np.max(np.abs(gx - Lx) / np.abs(Lx)) <= 1e-13
    with _csx.option("chol.forest", 0):
        S0 = cs.cs_schol(0, A)
        N0 = cs.cs_chol(A, S0)
        assert _chol_path(cs) == 0
    assert S0.parent == S.parent and S0.cp == S.cp
    g0p, g0i, g0x = _arr(N0.L)
    assert g0i.tolist() == Li.tolist() and np.max(np.abs(g0x - Lx) / np.abs(Lx)) <= 1e-13
    # the solve on that factor (general plan): cs_cholsol's answer
    b = synth.rhs(n, 1, 3)[:, 0]
    x = b.tolist()
    assert cs.cs_cholsol(0, A, x) is True
    ref = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b))
    assert np.max(np.abs(np.asarray(x) - ref)) <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("kind", ["tridiagonal", "arrow", "star", "random", "wide"])
def test_plan_partition_on_the_device_equals_the_hosts(cs, kind):
    """csx_cholsol_plan partitions a factor whose trees sit on consecutive columns without the host (block starts from a
    prefix maximum of the columns' last rows; one root to a block).  Same trees, same node lists, same programs as the
    host's partition_forest ("chol.forest" = 0): csx_cholsol_info equal and every solution bit for bit; "random" has
    blocks holding several trees (the device hands over to the host), "wide" a tree of 300 columns (level-scheduled)."""
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(4)
    sizes = list(rng.integers(1, 65, 120)) + [1, 64, 2]
    if kind == "wide":
        sizes = [300, 5, 40]
    n, Ap, Ai, Ax = _tree_blocks(sizes, 9, "arrow" if kind == "wide" else kind)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_schol(0, A)
    N = cs.cs_chol(A, S)
    gLp, gLi, gLx = _arr(N.L)
    L2 = cs.cs_pin(_host_cs(cs, n, n, gLp, gLi, gLx))
    k = 70
    B = synth.rhs(n, k, 1)
    out = []
    for forest in (1, 0):
        with _csx.option("chol.forest", forest):
            plan = _csx.new_handle()
            _csx.check(lib.csx_cholsol_plan(L2._dev.handle, None, plan))
            a, b, c = _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32()
            _csx.check(lib.csx_cholsol_info(plan, a, b, c))
            dB = cs.dvec(B)
            _csx.check(lib.csx_cholsol_solve(plan, dB.handle, k))
            out.append(((a.value, b.value, c.value), dB.numpy().copy()))
            _csx.free(plan)
    assert out[0][0] == out[1][0]
    if kind in ("tridiagonal", "arrow", "star"):
        assert out[0][0] == (1, len(sizes), max(sizes))
    assert out[0][1].tobytes() == out[1][1].tobytes()
    for r in (0, 33, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert out[0][1][:, r].tobytes() == ref.tobytes()


def test_forest_of_small_trees_refusals(cs):
    """A block wider than 64 columns, a pivot that is not positive, a column without a diagonal entry, an S of another
    matrix: the same outcomes as the general path."""
    import _csx
    # 40 + 40 columns joined by one entry: one block of 80 -> the general path
    n, Ap, Ai, Ax = _tree_blocks([40, 40, 7], 1, "random")
    cols = [(Ai[Ap[c]:Ap[c + 1]].tolist(), Ax[Ap[c]:Ap[c + 1]].tolist()) for c in range(n)]
    cols[79][0].insert(0, 2), cols[79][1].insert(0, 1e-3)
    cols[2][0].append(79), cols[2][1].append(1e-3)
    Ap2 = np.zeros(n + 1, np.int32)
    Ap2[1:] = np.cumsum([len(c[0]) for c in cols])
    Ai2 = np.concatenate([c[0] for c in cols]).astype(np.int32)
    Ax2 = np.concatenate([c[1] for c in cols])
    parent, cp = CO.schol(n, Ap2, Ai2)
    Lp, Li, Lx = CO.chol(n, Ap2, Ai2, Ax2, parent, cp)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap2, Ai2, Ax2))
    S = cs.cs_schol(0, A)
    N = cs.cs_chol(A, S)
    assert _chol_path(cs) == 0
    assert S.parent == parent.tolist() and S.cp == cp.tolist()
    assert np.max(np.abs(_arr(N.L)[2] - Lx) / np.abs(Lx)) <= 1e-13
    # not positive definite, in the middle of a block
    sizes = [9, 64, 30, 5]
    n, Ap, Ai, Ax = _tree_blocks(sizes, 2, "arrow")
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_schol(0, A)
    assert cs.cs_chol(A, S) is not None and _chol_path(cs) == 2
    c = 9 + 64 + 11
    Ax3 = Ax.copy()
    Ax3[Ap[c] + int(np.nonzero(Ai[Ap[c]:Ap[c + 1]] == c)[0][0])] = -2.0
    A3 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax3))
    assert cs.cs_chol(A3, S) is None
    with _csx.option("chol.forest", 0):
        assert cs.cs_chol(A3, S) is None
    # no diagonal entry in one column: the pivot is 0 (csparse.py:612)
    n4, Ap4, Ai4, Ax4 = _tree_blocks(sizes, 2, "arrow", missing_diagonal=c)
    par4, cp4 = CO.schol(n4, Ap4, Ai4)
    A4 = cs.cs_pin(_host_cs(cs, n4, n4, Ap4, Ai4, Ax4))
    S4 = cs.cs_schol(0, A4)
    assert S4.parent == par4.tolist() and S4.cp == cp4.tolist()
    assert cs.cs_chol(A4, S4) is None
    # an S that belongs to another matrix
    n5, Ap5, Ai5, Ax5 = _tree_blocks(sizes, 2, "tridiagonal")
    S5 = cs.cs_schol(0, cs.cs_pin(_host_cs(cs, n5, n5, Ap5, Ai5, Ax5)))
    outcomes = []
    for forest in (1, 0):
        with _csx.option("chol.forest", forest):
            try:
                outcomes.append(cs.cs_chol(A, S5) is None)
            except Exception as e:           # noqa: BLE001
                outcomes.append(type(e).__name__)
    assert outcomes[0] == outcomes[1]


def test_clique_forest_not_positive_definite_and_foreign_symbolic(cs):
    import _csx
    sizes = [5, 64, 9, 33]
    n, Ap, Ai, Ax = _blocks(sizes, 2)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    S = cs.cs_schol(0, A)
    assert cs.cs_chol(A, S) is not None
    # a negative pivot in the third block (csparse.py:612 -> None)
    c = 5 + 64 + 4
    Ax2 = Ax.copy()
    Ax2[Ap[c] + 4] = -1.0
    A2 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax2))
    assert cs.cs_chol(A2, S) is None
    with _csx.option("chol.clique", 0):
        assert cs.cs_chol(A2, S) is None
    # an S that belongs to another matrix: both paths refuse it the same way
    n3, Ap3, Ai3, Ax3 = _blocks([5, 64, 10, 32], 2)
    assert n3 == n
    S3 = cs.cs_schol(0, cs.cs_pin(_host_cs(cs, n, n, Ap3, Ai3, Ax3)))
    outcomes = []
    for clique in (1, 0):
        with _csx.option("chol.clique", clique):
            try:
                outcomes.append(cs.cs_chol(A, S3) is None)
            except Exception as e:           # noqa: BLE001
                outcomes.append(type(e).__name__)
    assert outcomes[0] == outcomes[1]


@pytest.mark.parametrize("bs", [8, 16, 32, 64])
def test_plan_cut_straight_out_of_the_factor(cs, bs):
    """The plan of a forest of equal dense blocks is made from L alone -- also from an L that went through the host (a
    received or re-uploaded factor) -- and solves with the bits of the general plan and of cs_lsolve + cs_ltsolve."""
    import _csx
    nblocks, k = 37, 70
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 99)
    n = nblocks * bs
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A, exact=True)
    assert F.info() == {"fused_local": True, "dense_block": bs, "matrix_cores": False, "trees": nblocks, "max_nodes": bs}
    gLp, gLi, gLx = _arr(F.L)
    B = synth.rhs(n, k, 0)
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    X = dB.numpy()
    for r in (0, 31, k - 1):
        ref = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
        assert X[:, r].tobytes() == ref.tobytes()
    with _csx.option("cholsol.dense_blocks", 0):     # the fused per-tree kernel on the same plan: programs made on demand
        dB1 = cs.dvec(B)
        assert F.solve(dB1) is True
        assert dB1.numpy().tobytes() == X.tobytes()
    with _csx.option("chol.clique", 0):              # the general plan (two triangular analyses, host partition)
        F0 = cs.cholsol_factor(A, exact=True)
        assert F0.info() == F.info()
        dB0 = cs.dvec(B)
        assert F0.solve(dB0) is True
        assert dB0.numpy().tobytes() == X.tobytes()
        F0r = cs.cholsol_factor(A, exact=False)
        dB0r = cs.dvec(B)
        assert F0r.solve(dB0r) is True
    Fr = cs.cholsol_factor(A, exact=False)            # matrix cores / FMA substitution from the clique plan's arrays
    dBr = cs.dvec(B)
    assert Fr.solve(dBr) is True
    assert dBr.numpy().tobytes() == dB0r.numpy().tobytes()
    assert np.max(np.abs(dBr.numpy() - X) / np.abs(X)) < 1e-13
    # a factor uploaded from host arrays
    L2 = cs.cs_pin(_host_cs(cs, n, n, gLp, gLi, gLx))
    plan = _csx.new_handle()
    _csx.check(_csx.lib().csx_cholsol_plan(L2._dev.handle, None, plan))
    a, b, c = _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32()
    _csx.check(_csx.lib().csx_cholsol_info(plan, a, b, c))
    assert (a.value, b.value, c.value) == (2, nblocks, bs)
    dB2 = cs.dvec(B)
    _csx.check(_csx.lib().csx_cholsol_solve(plan, dB2.handle, k))
    assert dB2.numpy().tobytes() == X.tobytes()
    _csx.free(plan)
    # a zero on the diagonal: ZeroDivisionError, as cs_lsolve raises
    gLx0 = gLx.copy()
    gLx0[gLp[n // 2]] = 0.0
    L3 = cs.cs_pin(_host_cs(cs, n, n, gLp, gLi, gLx0))
    plan = _csx.new_handle()
    _csx.check(_csx.lib().csx_cholsol_plan(L3._dev.handle, None, plan))
    assert _csx.lib().csx_cholsol_solve(plan, cs.dvec(B).handle, k) == _csx.EZEROPIVOT
    _csx.free(plan)


def test_default_order_follows_the_kind_of_right_hand_side(cs):
    """cholsol_factor(A) with exact=None: a list is solved in the reference's order (the reference's bits), a dvec block
    in the rounding-equal order (matrix cores here; inside 1e-10, not the same bits); exact=True / False pin one order."""
    nblocks, bs, k = 20, 32, 70
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 4)
    n = nblocks * bs
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    F = cs.cholsol_factor(A)
    gLp, gLi, gLx = _arr(F.L)
    B = synth.rhs(n, k, 0)
    ref0 = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, 0]))
    b = B[:, 0].tolist()
    assert F.solve(b) is True and np.asarray(b).tobytes() == ref0.tobytes()
    assert F.info()["matrix_cores"] is False
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    assert F.info()["matrix_cores"] is True                                  # the block went to the rounding-equal order
    X = dB.numpy()
    assert X[:, 0].tobytes() != ref0.tobytes() and np.max(np.abs(X[:, 0] - ref0) / np.abs(ref0)) < 1e-13
    b = B[:, 0].tolist()                                                      # and a list after it is exact again
    assert F.solve(b) is True and np.asarray(b).tobytes() == ref0.tobytes()
    Fe = cs.cholsol_factor(A, exact=True)
    dE = cs.dvec(B)
    assert Fe.solve(dE) is True and dE.numpy()[:, 0].tobytes() == ref0.tobytes()
    Fr = cs.cholsol_factor(A, exact=False)
    b = B[:, 0].tolist()
    assert Fr.solve(b) is True and np.asarray(b).tobytes() != ref0.tobytes()
    assert np.max(np.abs(np.asarray(b) - ref0) / np.abs(ref0)) < 1e-13
    b = B[:, 0].tolist()                                                      # the reference's own driver: always exact
    assert cs.cs_cholsol(0, A, b) is True and np.asarray(b).tobytes() == ref0.tobytes()


def test_the_finding_of_csx_schol_stays_on_the_matrix_until_it_is_invalidated(cs):
    """csx_schol leaves tree, counts and block list on the device matrix; the csx_chol that follows uses them instead of
    recognising the forest again.  Same factor with the finding, after csx_csc_invalidate (recognised anew) and for a
    matrix csx_schol never saw (csx_schol_host's S)."""
    import ctypes as C
    import _csx
    lib = _csx.lib()
    n, Ap, Ai, Ax = _blocks([64, 3, 17, 40, 1, 64], 8)
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)

    def factor(hA, par, cpp):
        hL = _csx.new_handle()
        _csx.check(lib.csx_chol(hA, _csx.pi(par), _csx.pi(cpp), None, hL))
        path = C.c_int32(-1)
        _csx.check(lib.csx_chol_info(path, None))
        x = np.empty(int(Lp[n]))
        _csx.check(lib.csx_csc_download(hL, None, None, _csx.pd(x)))
        _csx.free(hL)
        return path.value, x

    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    p1, x1 = factor(hA, parent, cp)
    par2, cp2 = np.empty(n, np.int32), np.empty(n + 1, np.int32)
    _csx.check(lib.csx_schol(hA, _csx.pi(par2), _csx.pi(cp2)))           # leaves its finding on the matrix
    assert par2.tolist() == parent.tolist() and cp2.tolist() == cp.tolist()
    p2, x2 = factor(hA, par2, cp2)
    _csx.check(lib.csx_csc_invalidate(hA))
    p3, x3 = factor(hA, par2, cp2)
    assert (p1, p2, p3) == (1, 1, 1)
    assert x1.tobytes() == Lx.tobytes() and x2.tobytes() == Lx.tobytes() and x3.tobytes() == Lx.tobytes()
    bad = cp2.copy()
    bad[3] += 1                                                             # an S that is not this matrix's: refused either way
    hL = _csx.new_handle()
    assert lib.csx_chol(hA, _csx.pi(par2), _csx.pi(bad), None, hL) == _csx.EINVAL
    _csx.free(hA)
