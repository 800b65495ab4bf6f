"""csx_cholsol_factor (round 5): cs_cholsol's factor sequence S = cs_schol(0, A); N = cs_chol(A, S) (csparse.py:636-639) and the
solve plan of :640-643 in ONE library call, S never leaving the device.  For a forest of equal dense blocks of 16 / 32 / 64
columns in the rounding-equal order the block kernel writes the matrix-core solve's operands beside L.x (path 3) and leaves
L.i to be made on demand.  Everything is compared with round 4's three calls (csx_schol + csx_chol + csx_cholsol_plan), which
are pinned to the plain-C oracle elsewhere (tests/test_gpu_cholclique.py), and with the oracle directly."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as CO
import synth
from test_gpu_cholclique import _arr, _blocks, _tree_blocks
from test_gpu_parity import _host_cs, cs  # noqa: F401
import tol as TOL

pytestmark = pytest.mark.gpu


def _download(h):
    import _csx
    lib = _csx.lib()
    m, n, nnz, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
    _csx.check(lib.csx_csc_info(h, m, n, nnz, hv))
    p = np.empty(n.value + 1, np.int32)
    i = np.empty(max(nnz.value, 1), np.int32)
    x = np.empty(max(nnz.value, 1), np.float64)
    _csx.check(lib.csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
    return p, i[:nnz.value], x[:nnz.value]


def _three_calls(A, n, exact):
    """round 4's flow through the C ABI: (L handle, plan handle, parent, cp)"""
    import _csx
    lib = _csx.lib()
    parent, cp = np.empty(max(n, 1), np.int32), np.empty(n + 1, np.int32)
    _csx.check(lib.csx_schol(A._dev.handle, _csx.pi(parent), _csx.pi(cp)))
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_chol(A._dev.handle, _csx.pi(parent), _csx.pi(cp), None, hL))
    _csx.check(lib.csx_cholsol_plan(hL, None, plan))
    if not exact:
        _csx.check(lib.csx_cholsol_set_order(plan, 0))
    return hL, plan, parent[:n], cp


def _fused(A, exact):
    import _csx
    lib = _csx.lib()
    hL, plan = _csx.new_handle(), _csx.new_handle()
    st = lib.csx_cholsol_factor(A._dev.handle, 1 if exact else 0, hL, plan)
    if st != _csx.OK:
        return st, None, None, None
    path = C.c_int32(-1)
    _csx.check(lib.csx_cholsol_factor_info(path, None, None, None))
    return st, hL, plan, path.value


def _solve(plan, B):
    import _csx
    import csparse
    dB = csparse.dvec(B)
    _csx.check(_csx.lib().csx_cholsol_solve(plan, dB.handle, B.shape[1]))
    return dB.numpy().copy()


def _info(plan):
    import _csx
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _csx.check(_csx.lib().csx_cholsol_info(plan, a, b, c))
    return a.value, b.value, c.value


CASES = {
    "equal64": lambda: (_blocks([64] * 41, 3), 3),
    "equal32": lambda: (_blocks([32] * 77, 4), 3),
    "equal16": lambda: (_blocks([16] * 130, 5), 3),
    "equal16_one_block": lambda: (_blocks([16], 6), 3),
    "equal8": lambda: (_blocks([8] * 90, 7), 1),
    "equal64_fill_inside": lambda: (_blocks([64] * 23, 8, 0.4), 3),        # entries missing inside the blocks: the kernel reads A.i
    "equal32_lower_shuffled": lambda: (_blocks([32] * 40, 9, 0.7, True), 3),
    "unequal": lambda: (_blocks(list(np.random.default_rng(1).integers(1, 65, 150)) + [64, 1, 16], 10), 1),
    "sparse_trees": lambda: (_tree_blocks(list(np.random.default_rng(2).integers(1, 65, 120)), 11, "arrow"), 2),
    "one_tree_general": lambda: (_tree_blocks([300, 5, 40], 12, "arrow"), 0),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("exact", [False, True])
def test_one_call_equals_the_three_calls(cs, case, exact):
    """L.p / L.i / L.x byte for byte; the solutions of the plan byte for byte, in the order the plan was made in and after
    switching it to the other one; the path csx_cholsol_factor_info reports."""
    import _csx
    lib = _csx.lib()
    (n, Ap, Ai, Ax), want_path = CASES[case]()
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    hL0, plan0, parent, cp = _three_calls(A, n, exact)
    st, hL, plan, path = _fused(A, exact)
    assert st == _csx.OK
    if want_path == 3 and exact:
        want_path = 1                 # the exact order needs the substitution programs: block kernel, then the plan from L.x
    assert path == want_path

    def same_info(in_exact_order):
        if case == "unequal" and in_exact_order:
            # the one call's plan solves cliques of unequal sizes exactly by PADDED SIZE CLASSES on the register-resident kernel
            # (path 2); the three calls' general plan by the fused per-tree kernel (path 1): same trees, same bits (below)
            assert _info(plan) == (2,) + _info(plan0)[1:] and _info(plan0)[0] == 1
        else:
            assert _info(plan) == _info(plan0)
    same_info(exact)
    k = 70
    B = synth.rhs(n, k, 5)
    X0 = _solve(plan0, B)
    X = _solve(plan, B)               # (before L is looked at: on path 3 the row indices do not exist yet)
    assert X.tobytes() == X0.tobytes()
    p0, i0, x0 = _download(hL0)
    p1, i1, x1 = _download(hL)
    assert p1.tobytes() == p0.tobytes() and i1.tobytes() == i0.tobytes() and x1.tobytes() == x0.tobytes()
    assert p1.tolist() == cp.tolist()
    # the analysis the call implies: cs_schol's tree is the first row below the diagonal of every column of L
    has = np.diff(p1) > 1
    par = np.full(n, -1, np.int32)
    par[has] = i1[p1[:-1][has] + 1]
    assert par.tolist() == parent.tolist()
    # the other order on the same plans
    for pl in (plan0, plan):
        _csx.check(lib.csx_cholsol_set_order(pl, 0 if exact else 1))
    same_info(not exact)
    Y0, Y = _solve(plan0, B), _solve(plan, B)
    assert Y.tobytes() == Y0.tobytes()
    # exact order: cs_lsolve + cs_ltsolve on this L (csparse.py:640-643), bit for bit; rounding-equal: 1e-10 componentwise
    Xe, Xr = (X, Y) if exact else (Y, X)
    for r in (0, 33, k - 1):
        ref = CO.ltsolve(n, p0, i0, x0, CO.lsolve(n, p0, i0, x0, B[:, r]))
        assert Xe[:, r].tobytes() == ref.tobytes()
        assert np.max(np.abs(Xr[:, r] - ref) / np.abs(ref)) <= TOL.X_RTOL
    # with the fused per-tree kernel ("cholsol.dense_blocks" = 0) the programs are cut out of L.x on demand
    with _csx.option("cholsol.dense_blocks", 0):
        assert _solve(plan, B).tobytes() == _solve(plan0, B).tobytes()
    for h in (plan, plan0, hL, hL0):
        _csx.free(h)


def test_not_positive_definite_and_the_guard(cs):
    import _csx
    lib = _csx.lib()
    n, Ap, Ai, Ax = _blocks([32] * 20, 13)
    # a negative pivot in the middle of the 8th block (csparse.py:612 -> None)
    c = 7 * 32 + 11
    Ax2 = Ax.copy()
    Ax2[Ap[c] + 11] = -3.0
    A2 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax2))
    for exact in (False, True):
        st, hL, plan, path = _fused(A2, exact)
        assert st == _csx.ENOTSPD
    assert cs.cs_cholsol(0, A2, [1.0] * n) is False
    assert cs.cholsol_factor(A2) is None
    # blocks scaled so that max|W| max|L| passes the guard's 1e3: the matrix-core operands are refused, the rounding-equal
    # order falls back to substitution out of L.x and still meets the tolerance
    Ax3 = Ax.copy()
    for b in range(20):
        sc = np.logspace(0, 3.6, 32)
        for cc in range(32):
            col = b * 32 + cc
            rows = Ai[Ap[col]:Ap[col + 1]] - b * 32
            Ax3[Ap[col]:Ap[col + 1]] *= sc[rows] * sc[cc]
    A3 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax3))
    st, hL, plan, path = _fused(A3, False)
    assert st == _csx.OK and path == 3
    g = C.c_double(0.0)
    _csx.check(lib.csx_cholsol_growth(plan, g))
    assert g.value > 1e3
    assert _info(plan)[0] == 2          # dense-block substitution, not the matrix cores
    hL0, plan0, _, _ = _three_calls(A3, n, False)
    g0 = C.c_double(0.0)
    _csx.check(lib.csx_cholsol_growth(plan0, g0))
    assert g0.value == g.value
    B = synth.rhs(n, 9, 2)
    X, X0 = _solve(plan, B), _solve(plan0, B)
    assert X.tobytes() == X0.tobytes()
    p0, i0, x0 = _download(hL)
    ref = CO.ltsolve(n, p0, i0, x0, CO.lsolve(n, p0, i0, x0, B[:, 4]))
    assert np.max(np.abs(X[:, 4] - ref) / np.abs(ref)) <= TOL.X_RTOL
    for h in (plan, plan0, hL, hL0):
        _csx.free(h)


def test_the_factor_handle_is_a_matrix_like_any_other(cs):
    """On path 3 L.i is written when a handle to L is first resolved: transpose, multiply, a triangular solve and a second
    plan on that handle see the complete matrix."""
    import _csx
    lib = _csx.lib()
    n, Ap, Ai, Ax = _blocks([16] * 25, 14)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    st, hL, plan, path = _fused(A, False)
    assert path == 3
    hT = _csx.new_handle()
    _csx.check(lib.csx_transpose(hL, 1, hT))              # the first use of the handle
    tp, ti, tx = _download(hT)
    p, i, x = _download(hL)
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    assert p.tolist() == Lp.tolist() and i.tolist() == Li.tolist() and x.tobytes() == Lx.tobytes()
    rp, ri, rx = CO.transpose(n, n, Lp, Li, Lx)
    assert tp.tolist() == rp.tolist() and ti.tolist() == ri.tolist() and tx.tobytes() == rx.tobytes()
    b = synth.rhs(n, 1, 8)[:, 0]
    L = cs.cs_spalloc(n, n, len(i), True, False)
    L.p, L.i, L.x = p.tolist(), i.tolist(), x.tolist()
    y = b.tolist()
    assert cs.cs_lsolve(L, y)
    assert np.asarray(y).tobytes() == CO.lsolve(n, Lp, Li, Lx, b).tobytes()
    for h in (hT, plan, hL):
        _csx.free(h)


def test_python_surface_goes_through_the_one_call(cs):
    """cholsol_factor(A) / cs_cholsol(0, A, b): lists exact (the reference's bits), blocks rounding-equal by default; the
    solver's `symbolic` is cs_schol(0, A)'s."""
    import _csx
    n, Ap, Ai, Ax = _blocks([64] * 12, 15)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    b = synth.rhs(n, 1, 4)[:, 0]
    ref = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b))
    x = b.tolist()
    assert cs.cs_cholsol(0, A, x) is True
    assert np.asarray(x).tobytes() == ref.tobytes()
    F = cs.cholsol_factor(A)
    path = C.c_int32(-1)
    _csx.check(_csx.lib().csx_cholsol_factor_info(path, None, None, None))
    assert path.value == 3
    assert F.info()["matrix_cores"] is True
    B = synth.rhs(n, 40, 4)
    dB = cs.dvec(B)
    assert F.solve(dB) is True
    assert np.max(np.abs(dB.numpy()[:, 0] - ref) / np.abs(ref)) <= TOL.X_RTOL
    y = b.tolist()
    assert F.solve(y) is True                               # a list: exact
    assert np.asarray(y).tobytes() == ref.tobytes()
    S = F.symbolic
    assert S.parent == parent.tolist() and S.cp == cp.tolist() and S.lnz == int(cp[n]) and S.pinv is None
    gp, gi, gx = _arr(F.L)
    assert gp.tolist() == Lp.tolist() and gi.tolist() == Li.tolist() and gx.tobytes() == Lx.tobytes()
    Fe = cs.cholsol_factor(A, exact=True)
    _csx.check(_csx.lib().csx_cholsol_factor_info(path, None, None, None))
    assert path.value == 1
    dBe = cs.dvec(B)
    assert Fe.solve(dBe) is True
    assert dBe.numpy()[:, 0].tobytes() == ref.tobytes()


@pytest.mark.parametrize("bs, density", [(64, 1.0), (32, 1.0), (16, 1.0), (64, 0.5), (8, 1.0)])
def test_rounding_equal_block_kernel(cs, bs, density):
    """"chol.exact" = 0 (opt-in): the block kernel with fused multiply-adds and refined reciprocal square roots.  L.p / L.i as
    always, L.x within 1e-13 (normwise) and 1e-12 (componentwise, entries above 1e-6 of the largest) of the default kernel's --
    which is bit-identical to the oracle -- through csx_chol and through csx_cholsol_factor; the solutions of the plan it made
    within 1e-10 of cs_lsolve + cs_ltsolve on the exact factor; a non-positive pivot still reported."""
    import _csx
    lib = _csx.lib()
    n, Ap, Ai, Ax = _blocks([bs] * 57, 31, density)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    S = cs.cs_schol(0, A)
    with _csx.option("chol.exact", 0):
        N = cs.cs_chol(A, S)
        st, hL, plan, path = _fused(A, False)
        assert st == _csx.OK
    gp, gi, gx = _arr(N.L)
    fp, fi, fx = _download(hL)
    for p_, i_, x_ in ((gp, gi, gx), (fp, fi, fx)):
        assert p_.tolist() == Lp.tolist() and i_.tolist() == Li.tolist()
        assert x_.tobytes() != Lx.tobytes()                              # it really is other arithmetic
        assert TOL.normwise(x_, Lx) <= 1e-13
        big = np.abs(Lx) > 1e-6 * np.abs(Lx).max()
        assert TOL.componentwise(x_[big], Lx[big]) <= 1e-12
    assert gx.tobytes() == fx.tobytes()                                  # one kernel behind both entries
    k = 70
    B = synth.rhs(n, k, 3)
    X = _solve(plan, B)
    for r in (0, 31, k - 1):
        y = CO.lsolve(n, Lp, Li, Lx, B[:, r])
        ref = CO.ltsolve(n, Lp, Li, Lx, y)
        assert TOL.componentwise(X[:, r], ref, TOL.cholsolve_terms(n, Lp, Li, Lx, y, ref)) <= TOL.X_RTOL
    # the default is untouched by the option having been used
    N1 = cs.cs_chol(A, S)
    assert _arr(N1.L)[2].tobytes() == Lx.tobytes()
    # not positive definite
    c = 5 * bs + bs // 2
    Ax2 = Ax.copy()
    Ax2[Ap[c] + int(np.nonzero(Ai[Ap[c]:Ap[c + 1]] == c)[0][0])] = -2.0
    A2 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax2))
    with _csx.option("chol.exact", 0):
        assert cs.cs_chol(A2, S) is None
        assert _fused(A2, False)[0] == _csx.ENOTSPD
    for h in (plan, hL):
        _csx.free(h)


def test_matrix_core_block_kernel_at_1m_rows(cs):
    """"chol.exact" = 0 on G-spd's shape at a fifth of its size (15 625 dense blocks of 64 columns): csx_cholsol_factor's L.x within
    1e-13 normwise of the default (bit-identical) kernel's, the fragments it wrote give the solutions of the default's plan to
    1e-12, A x = b holds, and two runs are bit-identical (stores through buffer resources, a kernel of 64 matrix instructions per
    block: checked at a size where every CU holds several blocks at once)."""
    import _csx
    lib = _csx.lib()
    nb, bs, k = 15625, 64, 64
    n = nb * bs
    hA = _csx.new_handle()
    _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
    hL0, plan0 = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hL0, plan0))
    outs = []
    with _csx.option("chol.exact", 0):
        for rep in range(2):
            hL, plan = _csx.new_handle(), _csx.new_handle()
            _csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan))
            outs.append((hL, plan))
    import csparse
    def lx(h):
        m_, n_, z, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
        _csx.check(lib.csx_csc_info(h, m_, n_, z, hv))
        x = np.empty(z.value)
        _csx.check(lib.csx_csc_download(h, None, None, _csx.pd(x)))
        return x
    x0, x1, x2 = lx(hL0), lx(outs[0][0]), lx(outs[1][0])
    assert x1.tobytes() == x2.tobytes() and x1.tobytes() != x0.tobytes()
    assert TOL.normwise(x1, x0) <= 1e-13
    sols = []
    for pl in (plan0, outs[0][1], outs[1][1]):
        hB = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
        _csx.check(lib.csx_cholsol_solve(pl, hB, k))
        a = np.empty(n * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(a), n * k))
        sols.append(a)
        _csx.free(hB)
    assert sols[1].tobytes() == sols[2].tobytes()
    assert float(np.max(np.abs(sols[1] - sols[0]) / np.abs(sols[0]))) <= 1e-12
    g = C.c_double(0.0), C.c_double(0.0)
    _csx.check(lib.csx_cholsol_growth(plan0, g[0]))
    _csx.check(lib.csx_cholsol_growth(outs[0][1], g[1]))
    assert abs(g[1].value - g[0].value) <= 1e-10 * g[0].value
    # A x = b on three columns (synth.gspd is the host twin of csx_gen_gspd)
    Ap, Ai, Ax = synth.gspd(nb, bs, 20240606)
    import scipy.sparse as sp
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    X = sols[1].reshape(n, k)
    for r in (0, 31, k - 1):
        b = 1.0 + (np.arange(n) + r) / float(n)
        assert float(np.max(np.abs(A @ X[:, r] - b))) <= 1e-12 * 128.0
    for h in (plan0, hL0, outs[0][1], outs[0][0], outs[1][1], outs[1][0], hA):
        _csx.free(h)


def test_matrix_core_block_kernel_on_unequal_cliques(cs):
    """"chol.exact" = 0 through csx_cholsol_factor on dense blocks of 1 .. 64 columns: the blocked factorisation per size class
    (a block padded with the identity to whole tiles), the size-class fragments written by it.  L.p / L.i the oracle's, L.x within
    1e-13 normwise, the plan on the matrix cores (path 5), solutions within 1e-10 of cs_lsolve + cs_ltsolve on the oracle's L;
    the exact order on the same plan bit-identical on ITS L; a non-positive pivot in a padded block reported."""
    import _csx
    lib = _csx.lib()
    rng = np.random.default_rng(17)
    sizes = list(rng.integers(1, 65, 220)) + [64, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63]
    n, Ap, Ai, Ax = _blocks(sizes, 41)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    with _csx.option("chol.exact", 0):
        st, hL, plan, path = _fused(A, False)
    assert st == _csx.OK and path == 1
    assert _info(plan)[0] == 5
    k = 70
    B = synth.rhs(n, k, 6)
    X = _solve(plan, B)
    fp, fi, fx = _download(hL)
    assert fp.tolist() == Lp.tolist() and fi.tolist() == Li.tolist()
    assert fx.tobytes() != Lx.tobytes() and TOL.normwise(fx, Lx) <= 1e-13
    big = np.abs(Lx) > 1e-6 * np.abs(Lx).max()
    assert TOL.componentwise(fx[big], Lx[big]) <= 1e-12
    for r in (0, 35, k - 1):
        y = CO.lsolve(n, Lp, Li, Lx, B[:, r])
        ref = CO.ltsolve(n, Lp, Li, Lx, y)
        assert TOL.componentwise(X[:, r], ref, TOL.cholsolve_terms(n, Lp, Li, Lx, y, ref)) <= TOL.X_RTOL
    _csx.check(lib.csx_cholsol_set_order(plan, 1))
    Xe = _solve(plan, B)
    for r in (0, k - 1):
        assert Xe[:, r].tobytes() == CO.ltsolve(n, fp, fi, fx, CO.lsolve(n, fp, fi, fx, B[:, r])).tobytes()
    # not positive definite, in a block that does not fill its last tile
    b = int(np.argmax(np.asarray(sizes) == 49))
    c = int(sum(sizes[:b])) + 40
    Ax2 = Ax.copy()
    Ax2[Ap[c] + int(np.nonzero(Ai[Ap[c]:Ap[c + 1]] == c)[0][0])] = -5.0
    A2 = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax2))
    with _csx.option("chol.exact", 0):
        assert _fused(A2, False)[0] == _csx.ENOTSPD
    for h in (plan, hL):
        _csx.free(h)


def test_edge_shapes_through_the_one_call(cs):
    """n = 0, n = 1, a diagonal matrix (every block one column), a matrix with no entries, a rectangular one, a missing diagonal:
    the drop-in's conventions (True / False / None) and the reference's arithmetic on each."""
    E = cs.cs_spalloc(0, 0, 0, True, False)
    E.p = [0]
    assert cs.cs_cholsol(0, E, []) is True
    F0 = cs.cholsol_factor(E)
    assert F0 is not None and F0.solve([]) is True
    one = cs.cs_spalloc(1, 1, 1, True, False)
    one.p, one.i, one.x = [0, 1], [0], [4.0]
    b = [6.0]
    assert cs.cs_cholsol(0, one, b) is True and b == [1.5]
    F1 = cs.cholsol_factor(one)
    assert F1.L.p == [0, 1] and F1.L.i[:1] == [0] and F1.L.x[:1] == [2.0]
    n = 300
    D = cs.cs_spalloc(n, n, n, True, False)
    D.p, D.i, D.x = list(range(n + 1)), list(range(n)), [float(2 + (j % 5)) for j in range(n)]
    rhs = [float(1 + j) for j in range(n)]
    want = [(rhs[j] / (D.x[j] ** 0.5)) / (D.x[j] ** 0.5) for j in range(n)]          # cs_lsolve then cs_ltsolve on L = sqrt(D)
    x = list(rhs)
    assert cs.cs_cholsol(0, D, x) is True and x == want
    FD = cs.cholsol_factor(D)
    Bk = cs.dvec(np.repeat(np.asarray(rhs)[:, None], 20, axis=1))
    assert FD.solve(Bk) is True
    assert np.max(np.abs(Bk.numpy().reshape(n, 20)[:, 7] - np.asarray(want)) / np.asarray(want)) <= 1e-14
    Z = cs.cs_spalloc(5, 5, 1, True, False)          # no entries at all: the first pivot is 0 -> not positive definite
    Z.p, Z.i, Z.x = [0] * 6, [0], [0.0]
    assert cs.cs_cholsol(0, Z, [1.0] * 5) is False and cs.cholsol_factor(Z) is None
    R = cs.cs_spalloc(3, 2, 2, True, False)
    R.p, R.i, R.x = [0, 1, 2], [0, 1], [1.0, 1.0]
    assert cs.cholsol_factor(R) is None
    M = cs.cs_spalloc(3, 3, 4, True, False)          # column 1 has no diagonal entry
    M.p, M.i, M.x = [0, 1, 2, 4], [0, 0, 1, 2], [4.0, 1.0, 1.0, 9.0]
    assert cs.cs_cholsol(0, M, [1.0, 1.0, 1.0]) is False
