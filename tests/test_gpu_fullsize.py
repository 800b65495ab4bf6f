"""BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot run
there in seconds): kernel-to-kernel agreement, linearity, transpose involution, A x = b residuals,
C 1 = A (A' 1).  Inputs are generated on the device (csx_gen_*), whose bit-exact agreement with the
host generators is covered by test_gpu_parity.py::test_generators_match_host_twins."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _vec(h, n, k=1):
    import csparse
    return csparse.dvec(n, k, _handle=h)


@pytest.fixture(scope="module")
def lib(cs):
    import _csx
    return _csx.lib()


def test_gaxpy_5m_modes_agree_and_are_linear(cs, lib):
    import _csx
    n, per_col = 5000000, 64
    hA = _csx.new_handle()
    _csx.check(lib.csx_gen_grand(n, per_col, 20240602, hA))
    hx = _csx.new_handle()
    _csx.check(lib.csx_gen_vec(n, 7, 0.5, 1.5, hx))
    x = _vec(hx, n)
    results = {}
    for name, mode in (("exact", cs.GAXPY_EXACT), ("wave", cs.GAXPY_WAVE), ("tiled", cs.GAXPY_TILED),
                       ("atomic", cs.GAXPY_ATOMIC), ("auto", cs.GAXPY_AUTO)):
        y = cs.dvec(n)
        _csx.check(lib.csx_gaxpy(hA, x.handle, y.handle, mode))
        results[name] = y.numpy()
    ref = results["exact"]  # reference summation order
    assert np.all(np.isfinite(ref)) and ref.min() > 0          # all-positive data: every row sum positive
    for name in ("wave", "tiled", "atomic", "auto"):
        assert np.max(np.abs(results[name] - ref) / ref) < 1e-12, name
    # total mass: 1' (A x) = sum_j x_j * colsum_j; column sums via y = A' 1 computed as gaxpy on the transpose
    hT = _csx.new_handle()
    _csx.check(lib.csx_transpose(hA, 1, hT))
    hone = _csx.new_handle()
    _csx.check(lib.csx_gen_vec(n, 1, 1.0, 1.0, hone))
    cols = cs.dvec(n)
    _csx.check(lib.csx_gaxpy(hT, hone, cols.handle, cs.GAXPY_WAVE))      # cols = A' 1
    lhs = float(np.sum(ref))
    rhs = float(np.dot(cols.numpy(), x.numpy()))
    assert abs(lhs - rhs) / abs(rhs) < 1e-12
    # linearity: A (2x) accumulated onto A x gives 3 A x (same kernel, y += semantics)
    y = cs.dvec(ref)
    x2 = cs.dvec(2.0 * x.numpy())
    _csx.check(lib.csx_gaxpy(hA, x2.handle, y.handle, cs.GAXPY_TILED))
    assert np.max(np.abs(y.numpy() - 3.0 * ref) / ref) < 1e-12
    # transpose is an involution on p and a permutation of the entries
    hTT = _csx.new_handle()
    _csx.check(lib.csx_transpose(hT, 1, hTT))
    p1, p2 = np.empty(n + 1, np.int32), np.empty(n + 1, np.int32)
    nnz = n * per_col
    i1, i2 = np.empty(nnz, np.int32), np.empty(nnz, np.int32)
    _csx.check(lib.csx_csc_download(hA, _csx.pi(p1), _csx.pi(i1), None))
    _csx.check(lib.csx_csc_download(hTT, _csx.pi(p2), _csx.pi(i2), None))
    assert (p1 == p2).all() and (i1 == i2).all()               # generator rows are ascending: (A')' == A
    tp = np.empty(n + 1, np.int32)
    _csx.check(lib.csx_csc_download(hT, _csx.pi(tp), None, None))
    assert tp[0] == 0 and tp[-1] == nnz and (np.diff(tp) >= 0).all()
    for h in (hA, hT, hTT, hone):
        _csx.free(h)


def test_gaxpy_5m_uniform_draw_the_benchmarked_matrix(cs, lib):
    """The exact workload of bench.py's `value`: csx_gen_grand_uniform(5M, 64, 20240601 + 1), x = csx_gen_vec(seed 7),
    the tiled plan with 3-byte keys.  The timed kernel's result against the reference-order kernel (bit-identical to
    csparse.py:1210-1212 on small sizes, tests/test_gpu_parity.py) at 1e-12, plus a sampled check of the device
    generator against its host twin at this size (columns picked across the range, rows and values bit for bit)."""
    import _csx
    import synth
    n, per_col, seed = 5000000, 64, 20240602
    hA = _csx.new_handle()
    _csx.check(lib.csx_gen_grand_uniform(n, per_col, seed, hA))
    # the generator at full size: 4 windows of 1000 columns against the host twin (the twin is pure per column)
    pA, iA, xA = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _csx.check(lib.csx_csc_ptrs(hA, pA, iA, xA))
    for j0 in (0, 1234567, 3999000, n - 1000):
        hw = _csx.new_handle()
        _csx.check(lib.csx_csc_col_block(hA, j0, 1000, hw))
        wp, wi, wx = np.empty(1001, np.int32), np.empty(64000, np.int32), np.empty(64000)
        _csx.check(lib.csx_csc_download(hw, _csx.pi(wp), _csx.pi(wi), _csx.pd(wx)))
        _csx.free(hw)
        ri, rx = synth.grand_uniform_columns(n, per_col, seed, j0, 1000)
        assert (wp == np.arange(1001) * 64).all() and (wi == ri).all() and wx.tobytes() == rx.tobytes(), j0
    hx = _csx.new_handle()
    _csx.check(lib.csx_gen_vec(n, 7, 0.5, 1.5, hx))
    _csx.check(lib.csx_gaxpy_prepare(hA, cs.GAXPY_TILED))
    kb = C.c_int(0)
    _csx.check(lib.csx_gaxpy_plan_info(hA, None, None, kb))
    assert kb.value == 3                                         # the plan the benchmark line reports (plan_key_bytes)
    yt, ye = cs.dvec(n), cs.dvec(n)
    _csx.check(lib.csx_gaxpy(hA, hx, yt.handle, cs.GAXPY_TILED))
    _csx.check(lib.csx_gaxpy(hA, hx, ye.handle, cs.GAXPY_EXACT))
    a, e = yt.numpy(), ye.numpy()
    assert np.all(np.isfinite(e)) and e.min() >= 0
    nz = e > 0                                                   # a row can be empty under the uniform draw
    assert (a[~nz] == 0).all()
    assert np.max(np.abs(a[nz] - e[nz]) / e[nz]) < 1e-12
    # the 4-byte-key plan of the same matrix agrees too (and really is another plan)
    with _csx.option("gaxpy.keys24", 0):
        _csx.check(lib.csx_csc_invalidate(hA))
        _csx.check(lib.csx_gaxpy_prepare(hA, cs.GAXPY_TILED))
        _csx.check(lib.csx_gaxpy_plan_info(hA, None, None, kb))
        assert kb.value == 4
        y4 = cs.dvec(n)
        _csx.check(lib.csx_gaxpy(hA, hx, y4.handle, cs.GAXPY_TILED))
        b = y4.numpy()
        assert np.max(np.abs(b[nz] - e[nz]) / e[nz]) < 1e-12
    for h in (hA, hx):
        _csx.free(h)


def test_clique_forest_at_5m_equals_the_general_path_bit_for_bit(cs, lib):
    """Config 5's matrix at full size: csx_schol and csx_chol through the clique-forest path (csx_cholclique.hip) against the
    general path ("chol.clique" = 0: pattern machine, k_chol_dense_trees).  Both keep the reference's operation order on
    dense blocks, and both are bit-identical to the plain-C port at sizes it reaches (tests/test_gpu_cholesky.py,
    tests/test_gpu_cholclique.py): at 5M rows parent, cp, L.p, L.i and L.x must be EQUAL, byte for byte (digests), and the
    solve plans must give the same bits."""
    import hashlib
    import _csx
    nb, bs, k = 78125, 64, 8
    n = nb * bs
    hA = _csx.new_handle()
    _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
    lnz = nb * bs * (bs + 1) // 2

    def run():
        parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
        _csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)))
        hL = _csx.new_handle()
        _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
        path = C.c_int32(-1)
        _csx.check(lib.csx_chol_info(path, None))
        Lp, Li, Lx = np.empty(n + 1, np.int32), np.empty(lnz, np.int32), np.empty(lnz)
        _csx.check(lib.csx_csc_download(hL, _csx.pi(Lp), _csx.pi(Li), _csx.pd(Lx)))
        plan, hB = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_cholsol_plan(hL, None, plan))
        _csx.check(lib.csx_gen_rhs(n, k, 3, hB))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        X = np.empty(n * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(X), n * k))
        dig = [hashlib.sha256(a.tobytes()).hexdigest() for a in (parent, cp, Lp, Li, Lx, X)]
        for h in (plan, hB, hL):
            _csx.free(h)
        return path.value, dig

    p1, d1 = run()
    with _csx.option("chol.clique", 0):
        _csx.check(lib.csx_csc_invalidate(hA))               # drop the finding csx_schol left on the matrix
        p0, d0 = run()
    _csx.free(hA)
    assert (p1, p0) == (1, 0)
    assert d1 == d0


def test_forest_of_small_sparse_trees_at_4_8m_rows(cs, lib):
    """200 000 blocks of 24 columns, tridiagonal plus a full last row (small elimination trees that are chains but no cliques):
    csx_schol / csx_chol through the sparse-forest path (symbolic elimination on row masks, the block kernel with a compacted
    store; csx_chol_info path 2) and the plan partitioned on the device, against the general machines ("chol.forest" = 0):
    parent, cp, L.p, L.i EQUAL (digests); L.x of both equal to rounding, and -- chains are eliminated in the reference's order --
    the forest path's first and last 2 000 columns byte for byte the plain-C port's (blocks are independent, so the port runs on
    slices); the exact-order solutions of the two plans' own factors satisfy A x = b to 1e-12."""
    import hashlib
    import _csx
    import c_oracle as CO
    nb, bs, k = 200000, 24, 8
    n = nb * bs
    cols = []
    for c in range(bs):
        rows = {c, bs - 1} | ({c - 1} if c > 0 else set()) | ({c + 1} if c + 1 < bs else set())
        cols.append(sorted(range(bs)) if c == bs - 1 else sorted(rows))
    bi = np.concatenate([np.asarray(r, np.int64) for r in cols])
    bp = np.concatenate([[0], np.cumsum([len(r) for r in cols])])
    rng = np.random.default_rng(5)
    Ap = np.concatenate([(np.arange(nb)[:, None] * bp[-1] + bp[None, :-1]).reshape(-1), [nb * bp[-1]]]).astype(np.int32)
    Ai = (bi[None, :] + (np.arange(nb) * bs)[:, None]).reshape(-1).astype(np.int32)
    # symmetric values: entry (r, c) of block b = -u(b, min, max) / (1 + |r - c|), the diagonal 8 + u
    colof = np.repeat(np.arange(bs), np.diff(bp))
    lo, hi = np.minimum(bi, colof), np.maximum(bi, colof)
    U = rng.uniform(0.5, 1.0, size=(nb, bs))
    Ax = np.where(lo[None, :] == hi[None, :], 8.0 + U[:, lo], -U[:, lo] * U[:, hi] / (1.0 + (hi - lo)[None, :])).reshape(-1)
    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))

    def run():
        parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
        _csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)))
        hL = _csx.new_handle()
        _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
        path = C.c_int32(-1)
        _csx.check(lib.csx_chol_info(path, None))
        lnz = int(cp[n])
        Lp, Li, Lx = np.empty(n + 1, np.int32), np.empty(lnz, np.int32), np.empty(lnz)
        _csx.check(lib.csx_csc_download(hL, _csx.pi(Lp), _csx.pi(Li), _csx.pd(Lx)))
        plan, hB = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_cholsol_plan(hL, None, plan))
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _csx.check(lib.csx_cholsol_info(plan, a, b, c))
        _csx.check(lib.csx_gen_rhs(n, k, 3, hB))
        B0 = np.empty(n * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(B0), n * k))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        X = np.empty(n * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(X), n * k))
        dig = [hashlib.sha256(v.tobytes()).hexdigest() for v in (parent, cp, Lp, Li)]
        for h in (plan, hB, hL):
            _csx.free(h)
        return path.value, (a.value, b.value, c.value), dig, (parent, cp, Lp, Li, Lx), B0.reshape(n, k), X.reshape(n, k)

    p1, i1, d1, f1, B1, X1 = run()
    with _csx.option("chol.forest", 0):
        _csx.check(lib.csx_csc_invalidate(hA))
        p0, i0, d0, f0, B0, X0 = run()
    _csx.free(hA)
    assert (p1, p0) == (2, 0)
    assert i1 == i0 == (1, nb, bs)
    assert d1 == d0
    assert np.max(np.abs(f1[4] - f0[4]) / np.abs(f0[4])) <= 1e-13
    # the port on the first and the last 2 000 columns
    for c0, c1 in ((0, 2016), (n - 2016, n)):
        sp = Ap[c0:c1 + 1] - Ap[c0]
        si, sx = Ai[Ap[c0]:Ap[c1]] - c0, Ax[Ap[c0]:Ap[c1]]
        par, cpp = CO.schol(c1 - c0, sp, si)
        Lp, Li, Lx = CO.chol(c1 - c0, sp, si, sx, par, cpp)
        got = f1[4][f1[2][c0]:f1[2][c1]]
        assert (f1[3][f1[2][c0]:f1[2][c1]] - c0).tolist() == Li.tolist()
        assert got.tobytes() == Lx.tobytes()
    # residual of both solves (A is symmetric: a column of A is a row)
    import scipy.sparse as sps
    A = sps.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    for B, X in ((B1, X1), (B0, X0)):
        R = A @ X - B
        assert np.max(np.abs(R)) <= 1e-12 * np.max(np.abs(B))


def test_cholsol_5m_block_spd_residual(cs, lib):
    import _csx
    nb, bs, k = 78125, 64, 128
    n = nb * bs
    hA = _csx.new_handle()
    _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
    p = np.empty(n + 1, np.int32)
    i = np.empty(n * bs, np.int32)
    _csx.check(lib.csx_csc_download(hA, _csx.pi(p), _csx.pi(i), None))
    parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
    _csx.check(lib.csx_schol_host(n, _csx.pi(p), _csx.pi(i), _csx.pi(parent), _csx.pi(cp)))
    assert int(cp[n]) == nb * bs * (bs + 1) // 2                # no fill: lnz = 78125 * 2080
    hL = _csx.new_handle()
    _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
    plan = _csx.new_handle()
    _csx.check(lib.csx_cholsol_plan(hL, None, plan))
    path, trees, mx = C.c_int32(), C.c_int32(), C.c_int32()
    _csx.check(lib.csx_cholsol_info(plan, path, trees, mx))
    assert (path.value, trees.value, mx.value) == (2, nb, bs)   # default order: dense-block substitution, the reference's bits
    _csx.check(lib.csx_cholsol_set_order(plan, 0))              # the benchmark's leg: rounding-equal order
    _csx.check(lib.csx_cholsol_info(plan, path, trees, mx))
    assert (path.value, trees.value, mx.value) == (3, nb, bs)   # dense blocks on the matrix cores
    hB = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
    dB = _vec(hB, n, k)   # owns hB from here on (freed with the object)
    B0 = dB.numpy().copy()
    _csx.check(lib.csx_cholsol_solve(plan, hB, k))
    X = dB.numpy()
    # residual of three columns against the full symmetric matrix: A x - b
    for r in (0, 63, 127):
        xr = cs.dvec(np.ascontiguousarray(X[:, r]))
        res = cs.dvec(-B0[:, r])
        _csx.check(lib.csx_gaxpy(hA, xr.handle, res.handle, cs.GAXPY_WAVE))
        assert np.max(np.abs(res.numpy())) < 1e-12 * np.max(np.abs(B0[:, r])) * bs
    # the reference-order fused kernel agrees with the dense-block kernel to rounding
    with _csx.option("cholsol.dense_blocks", 0):
        hB2 = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, 0, hB2))
        dB2 = _vec(hB2, n, k)
        _csx.check(lib.csx_cholsol_solve(plan, hB2, k))
        X2 = dB2.numpy()
    assert np.max(np.abs(X2 - X) / np.abs(X2)) < 1e-12
    for h in (plan, hL, hA):
        _csx.free(h)


@pytest.mark.parametrize("draw", ["uniform", "stratified"])
def test_multiply_1m_identity(cs, lib, draw):
    """Config 4 at full size.  "uniform" is the matrix bench_configs.py times (csx_gen_grand_uniform, seed 20240605: 32
    distinct uniform rows per column, ragged rows, more hash collisions); "stratified" round 1's draw."""
    import _csx
    n, per_col = 1000000, 32
    hA = _csx.new_handle()
    _csx.check((lib.csx_gen_grand_uniform if draw == "uniform" else lib.csx_gen_grand)(n, per_col, 20240605, hA))
    hB = _csx.new_handle()
    _csx.check(lib.csx_transpose(hA, 1, hB))
    hC = _csx.new_handle()
    _csx.check(lib.csx_multiply(hA, hB, hC))
    m_, n_, nnzC, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
    _csx.check(lib.csx_csc_info(hC, m_, n_, nnzC, hv))
    assert (m_.value, n_.value, hv.value) == (n, n, 1)
    assert 0.9e9 < nnzC.value <= n * per_col * per_col           # <= number of products, few collisions
    cp = np.empty(n + 1, np.int32)
    _csx.check(lib.csx_csc_download(hC, _csx.pi(cp), None, None))
    assert cp[0] == 0 and cp[-1] == nnzC.value and (np.diff(cp) > 0).all()
    hone = _csx.new_handle()
    _csx.check(lib.csx_gen_vec(n, 1, 1.0, 1.0, hone))
    t1, t2, t3 = cs.dvec(n), cs.dvec(n), cs.dvec(n)
    _csx.check(lib.csx_gaxpy(hB, hone, t1.handle, cs.GAXPY_WAVE))        # A' 1
    _csx.check(lib.csx_gaxpy(hA, t1.handle, t2.handle, cs.GAXPY_WAVE))   # A (A' 1)
    _csx.check(lib.csx_gaxpy(hC, hone, t3.handle, cs.GAXPY_ATOMIC))      # C 1
    a, c = t2.numpy(), t3.numpy()
    assert np.max(np.abs(a - c) / np.abs(a)) < 1e-12
    for h in (hA, hB, hC, hone):
        _csx.free(h)


def test_forest_of_unequal_cliques_at_1m_rows_rounding_equal_equals_exact_everywhere(cs, lib):
    """csx_trimfma.hip at scale: a forest of 28 000 cliques of 8 .. 64 columns (1M rows), 128 right-hand sides: EVERY entry of the
    rounding-equal solve (path 5: trees made dense by size class on the matrix cores) within 1e-12 of the exact order's, and two
    runs of it bit-identical.  (This size is what it takes: a store hazard -- a 16-byte buffer store with a scalar offset whose data
    registers the next vector instruction overwrites, which the compiler does not pad -- corrupted the seventh digit of about one
    block in a hundred at 1M and 5M rows and never at the few thousand rows of tests/test_gpu_trimfma.py.)"""
    import _csx
    import synth
    n, Ap, Ai, Ax, sizes = synth.ragged_cliques(1000000, 8, 64, 20240605)
    k = 128
    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan))
    path = C.c_int32(-1)
    _csx.check(lib.csx_cholsol_info(plan, path, None, None))
    assert path.value == 5
    sols = []
    for exact in (1, 0, 0):
        _csx.check(lib.csx_cholsol_set_order(plan, exact))
        hB = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        x = np.empty(n * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(x), n * k))
        sols.append(x)
        _csx.free(hB)
    assert sols[1].tobytes() == sols[2].tobytes()
    assert float(np.max(np.abs(sols[1] - sols[0]) / np.abs(sols[0]))) <= 1e-12
    # and the answer is a solution: A x = b for three columns (A symmetric, full storage)
    import scipy.sparse as sp
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    X = sols[1].reshape(n, k)
    for r in (0, 77, k - 1):
        b = 1.0 + (np.arange(n) + r) / float(n)
        assert float(np.max(np.abs(A @ X[:, r] - b))) <= 1e-12 * 64.0
    for h in (plan, hL, hA):
        _csx.free(h)
