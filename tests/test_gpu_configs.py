"""The BASELINE.json configs that are not the headline, at their stated sizes, in the driver-run suite:
config 2 (cs_gaxpy on the symmetric-expanded bcsstk16, 290 378 entries) against vectors the UNMODIFIED
reference produced (tests/golden/config2_bcsstk16.npz, oracle/gen_golden.py), and config 3 (cs_lusol on W,
the west0067 tiling, n = 100 031) against the plain-C oracle.  Plus csx_cumsum called directly
(csparse.py:767-784) and csx_csc_invalidate after an in-place change of A.x through csx_csc_ptrs."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as CO
import synth
from conftest import golden
from test_gpu_parity import RTOL, _host_cs, abs_terms, cs, rel_err  # noqa: F401

pytestmark = pytest.mark.gpu


def _config2(cs):
    g, v = golden("bcsstk16"), golden("config2_bcsstk16")
    p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
    assert len(i) == 290378 and len(p) == 4885
    return _host_cs(cs, 4884, 4884, p, i, x), v


def test_config2_list_call_bit_exact_vs_reference(cs):
    A, v = _config2(cs)
    x, y = v["x"].tolist(), v["y0"].tolist()
    assert cs.cs_gaxpy(A, x, y) is True
    assert np.asarray(y).tobytes() == v["y"].tobytes()          # the unmodified reference's bits
    assert x == v["x"].tolist()


@pytest.mark.parametrize("mode", ["EXACT", "WAVE", "ATOMIC", "TILED", "AUTO"])
def test_config2_device_modes(cs, mode):
    A, v = _config2(cs)
    cs.cs_pin(A)
    dx, dy = cs.dvec(v["x"]), cs.dvec(v["y0"])
    assert cs.cs_gaxpy(A, dx, dy, getattr(cs, "GAXPY_" + mode)) is True
    got = dy.numpy()
    if mode == "EXACT":
        assert got.tobytes() == v["y"].tobytes()
    scale = abs_terms(A, v["x"]) + np.abs(v["y0"])
    assert rel_err(got, v["y"], scale) < RTOL


def _w_matrix(nb):
    """W of SURVEY 8d: block-diagonal tiling of the drop-tol'd west0067 pattern (csparse_test.py:633), block b
    scaled by 1 + 1e-3 u_b.  Same construction as bench_configs.config3."""
    g = golden("west0067")
    bp, bi, bx = g["C_p"].astype(np.int64), g["C_i"].astype(np.int64), g["C_x"]
    bs = 67
    u = synth.vec(nb, 20240604, 0.0, 1.0)
    Ai = (bi[None, :] + (np.arange(nb) * bs)[:, None]).reshape(-1).astype(np.int32)
    Ax = (bx[None, :] * (1.0 + 1e-3 * u)[:, None]).reshape(-1)
    Ap = np.concatenate([[0], np.cumsum(np.tile(np.diff(bp), nb))]).astype(np.int32)
    return nb * bs, Ap, Ai, Ax


def test_config3_lusol_on_W_full_size(cs):
    """cs_lusol's sequence on W (n = 100 031): host LU, device cs_lsolve + cs_usolve bit-identical to the plain-C
    oracle for 1 and 64 right-hand sides; residual of the solution against A."""
    n, Ap, Ai, Ax = _w_matrix(1493)
    assert n == 100031
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    N = cs.cs_lu(A, cs.cs_sqr(0, A, False), 1.0)
    assert N is not None
    L, U = cs.cs_pin(N.L), cs.cs_pin(N.U)
    b = 1.0 + np.arange(n) / n
    pb = np.empty(n)
    pb[np.asarray(N.pinv)] = b                                   # x = b(p)   (cs_ipvec)
    Lp, Li, Lx = (np.asarray(v) for v in (L.p, L.i, L.x))
    Up, Ui, Ux = (np.asarray(v) for v in (U.p, U.i, U.x))
    ref_y = CO.lsolve(n, Lp.astype(np.int32), Li.astype(np.int32), Lx, pb)
    ref_x = CO.usolve(n, Up.astype(np.int32), Ui.astype(np.int32), Ux, ref_y)
    for k in (1, 64):
        X = cs.dvec(np.repeat(pb[:, None], k, axis=1) if k > 1 else pb)
        assert cs.cs_lsolve(L, X) is True
        Y = X.numpy().reshape(n, -1)
        for r in (0, k - 1):
            assert Y[:, r].tobytes() == ref_y.tobytes(), (k, r)
        assert cs.cs_usolve(U, X) is True
        got = X.numpy().reshape(n, -1)
        for r in range(k):
            assert got[:, r].tobytes() == ref_x.tobytes(), (k, r)
    # the list-based driver end to end (q is the identity for order 0)
    bl = b.tolist()
    assert cs.cs_lusol(0, A, bl, 1.0) is True
    assert np.asarray(bl).tobytes() == ref_x.tobytes()
    # the batched form (factor once; permute, L, U, permute on the device): 70 right-hand sides, scaled copies of b, every column
    # against the plain-C oracle's solves on the same factors
    F = cs.lusol_factor(A, 0, 1.0, exact=True)
    scales = 1.0 + 0.5 * np.arange(70)
    dB = cs.dvec(np.ascontiguousarray(b[:, None] * scales[None, :]))
    assert F.solve(dB) is True
    Xb = dB.numpy().reshape(n, 70)
    FL, FU = F.factors.L, F.factors.U
    fLp, fLi, fLx = (np.asarray(v) for v in (FL.p, FL.i, FL.x))
    fUp, fUi, fUx = (np.asarray(v) for v in (FU.p, FU.i, FU.x))
    fpinv = np.asarray(F.factors.pinv)
    for r in (0, 33, 69):
        pbr = np.empty(n)
        pbr[fpinv] = b * scales[r]
        want = CO.usolve(n, fUp.astype(np.int32), fUi.astype(np.int32), fUx,
                         CO.lsolve(n, fLp.astype(np.int32), fLi.astype(np.int32), fLx, pbr))
        assert Xb[:, r].tobytes() == want.tobytes(), r
    # the default of lusol_factor solves a device block in the rounding-equal order (round 5: W's components dense on the matrix
    # cores, tests/test_gpu_trimfma.py): the same columns within BASELINE's 1e-10
    Fd = cs.lusol_factor(A, 0, 1.0)
    dBd = cs.dvec(np.ascontiguousarray(b[:, None] * scales[None, :]))
    assert Fd.solve(dBd) is True
    Xd = dBd.numpy().reshape(n, 70)
    assert all(v["matrix_cores"] for v in Fd.info().values())
    for r in (0, 33, 69):
        assert np.max(np.abs(Xd[:, r] - Xb[:, r]) / np.abs(Xb[:, r])) <= 1e-10, r
    res = CO.gaxpy(n, n, Ap, Ai, Ax, ref_x, -b)
    norm1 = float(np.max(np.add.reduceat(np.abs(Ax), Ap[:-1])))
    assert np.max(np.abs(res)) <= 1e-12 * (norm1 * np.max(np.abs(ref_x)) + np.max(np.abs(b)))


def test_cumsum_on_device_vs_reference_fixture(cs):
    """csx_cumsum called directly (csparse.py:767-784): p, the overwritten c and the returned total, against the
    reference's own answers (tests/golden/synthetic_20240601.npz), then at a size that needs several scan blocks."""
    import _csx
    lib = _csx.lib()
    g = golden("synthetic_20240601")
    cases = [(g["cumsum_c"].astype(np.int32), g["cumsum_p"].astype(np.int32), g["cumsum_c_out"].astype(np.int32),
              int(g["cumsum_ret"][0]))]
    rng = np.random.default_rng(3)
    for n in (1, 2, 255, 256, 257, 100000, 3000001):
        c = rng.integers(0, 9, size=n).astype(np.int32)
        p = np.concatenate([[0], np.cumsum(c)]).astype(np.int32)
        cases.append((c, p, p[:-1].copy(), int(p[-1])))
    cases.append((np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32), 0))
    for c, p_ref, c_ref, total_ref in cases:
        n = len(c)
        hc, hp = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_ivec_upload(_csx.pi(np.ascontiguousarray(c)) if n else None, n, hc))
        _csx.check(lib.csx_ivec_upload(_csx.pi(np.full(n + 1, -7, np.int32)), n + 1, hp))
        total = C.c_int64(-1)
        _csx.check(lib.csx_cumsum(hp, hc, n, total))
        p_out, c_out = np.empty(n + 1, np.int32), np.empty(max(n, 1), np.int32)
        _csx.check(lib.csx_ivec_download(hp, _csx.pi(p_out), n + 1))
        if n:
            _csx.check(lib.csx_ivec_download(hc, _csx.pi(c_out), n))
        assert total.value == total_ref and p_out.tolist() == p_ref.tolist(), n
        assert c_out[:n].tolist() == c_ref.tolist(), n
        _csx.free(hc)
        _csx.free(hp)


def test_in_place_change_of_values_needs_invalidate(cs):
    """The SpMV plans cached on a matrix copy its values: after A.x changed through csx_csc_ptrs every planned
    mode is stale until csx_csc_invalidate, and agrees with the oracle afterwards (ATOMIC reads A directly)."""
    import _csx
    lib = _csx.lib()
    n, per_col = 20000, 16
    Ap, Ai, Ax = synth.grand(n, per_col, 91)
    x = synth.vec(n, 5, 0.5, 1.5)
    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    dx = cs.dvec(x)
    modes = (cs.GAXPY_EXACT, cs.GAXPY_WAVE, cs.GAXPY_TILED)
    ref1 = CO.gaxpy(n, n, Ap, Ai, Ax, x, np.zeros(n))
    for m in modes:
        dy = cs.dvec(n)
        _csx.check(lib.csx_gaxpy(hA, dx.handle, dy.handle, m))
        assert rel_err(dy.numpy(), ref1) < RTOL
    # overwrite A.x in place on the device
    dp, di, dxp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _csx.check(lib.csx_csc_ptrs(hA, dp, di, dxp))
    Ax2 = Ax * synth.vec(len(Ax), 17, 2.0, 3.0)
    hv = _csx.new_handle()
    _csx.check(lib.csx_vec_wrap(dxp, len(Ax2), hv))
    _csx.check(lib.csx_vec_write(hv, _csx.pd(Ax2), len(Ax2)))
    _csx.free(hv)
    ref2 = CO.gaxpy(n, n, Ap, Ai, Ax2, x, np.zeros(n))
    dy = cs.dvec(n)
    _csx.check(lib.csx_gaxpy(hA, dx.handle, dy.handle, cs.GAXPY_ATOMIC))
    assert rel_err(dy.numpy(), ref2) < RTOL                       # reads the live arrays
    dy = cs.dvec(n)
    _csx.check(lib.csx_gaxpy(hA, dx.handle, dy.handle, cs.GAXPY_WAVE))
    assert rel_err(dy.numpy(), ref1) < RTOL                       # documented: the cached plan is stale
    _csx.check(lib.csx_csc_invalidate(hA))
    for m in modes + (cs.GAXPY_AUTO,):
        dy = cs.dvec(n)
        _csx.check(lib.csx_gaxpy(hA, dx.handle, dy.handle, m))
        assert rel_err(dy.numpy(), ref2) < RTOL, m
    # a freed handle is dead, and a new object in the same slot does not answer to it
    old = C.c_uint64(hA.value)
    _csx.free(hA)
    hB = _csx.new_handle()
    _csx.check(lib.csx_vec_alloc(8, hB))
    assert lib.csx_csc_invalidate(old) == _csx.EINVAL and lib.csx_free(old) == _csx.EINVAL
    assert (hB.value & 0xffffffff) == (old.value & 0xffffffff) and hB.value != old.value
    _csx.free(hB)


def test_results_on_pinned_inputs_stop_being_device_backed_once_read(cs):
    """C = cs_transpose(pinned A) lives on the device until its lists are read; after that the lists are the only
    copy, so the reference's idiom C.x[k] = v is seen by the next call."""
    n, per_col = 3000, 8
    Ap, Ai, Ax = synth.grand(n, per_col, 7)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    T = cs.cs_transpose(A, True)
    assert T._lazy and T._dev is not None
    x = synth.vec(n, 2, 0.5, 1.5)
    y = cs.dvec(n)
    assert cs.cs_gaxpy(T, cs.dvec(x), y, cs.GAXPY_EXACT)          # still lazy: no download happened
    assert T._lazy
    v = T.x[5]                                                    # the caller looks at the values ...
    assert not T._lazy and T._dev is None
    T.x[5] = v + 100.0                                            # ... and edits one in place
    Tp, Ti, Tx = np.asarray(T.p, np.int32), np.asarray(T.i, np.int32), np.asarray(T.x)
    ref = CO.gaxpy(n, n, Tp, Ti, Tx, x, np.zeros(n))
    yl = [0.0] * n
    assert cs.cs_gaxpy(T, x.tolist(), yl)
    assert np.asarray(yl).tobytes() == ref.tobytes()


def test_connected_problem_section_of_the_bench_runs_and_solves(cs):
    """bench_configs.cholsol_connected (bench.py carries it under other_configs): bcsstk16 and a grid Laplacian, natural
    order and order 1, one-shot call and both solve orders -- every residual small, the factor sizes as the symbolic
    analysis says.  (A smaller grid than the bench's, same code path.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench_configs as bc
    keep = bc.SKIP_CPU
    bc.SKIP_CPU = True
    try:
        out = bc.cholsol_connected(grid=90)["results"]
    finally:
        bc.SKIP_CPU = keep
    for name, res in out.items():
        for order in ("order_0", "order_1"):
            r = res[order]
            assert r["residual_inf"] is not None and r["residual_inf"] < 1e-9, (name, order, r)
            assert r["lnz"] > res["n"] and r["cs_chol_ms"] > 0
    assert out["bcsstk16"]["order_0"]["lnz"] == 610800
