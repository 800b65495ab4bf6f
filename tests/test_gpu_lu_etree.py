"""cs_lu of ONE connected matrix on the device (csx_lu_etree: columns scheduled by the column elimination tree, a lane per
column running the host code's loop) against csx_lu_host: L, U and pinv bit for bit.  The host code itself is pinned by
the unmodified reference's cs_lusol answers (tests/test_host_symbolic.py, tests/test_gpu_cholesky.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from test_gpu_parity import _host_cs, cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _host_lu(n, Ap, Ai, Ax, tol):
    import _csx
    out = [C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)(),
           C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()]
    pinv = np.empty(n, np.int32)
    lib = _csx.load()
    st = lib.csx_lu_host(n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), float(tol), *[C.byref(o) for o in out], _csx.pi(pinv))
    if st == _csx.ENOTSPD:
        return None
    _csx.check(st)
    Lp = np.ctypeslib.as_array(out[0], shape=(n + 1,)).copy()
    Up = np.ctypeslib.as_array(out[3], shape=(n + 1,)).copy()
    res = (Lp, np.ctypeslib.as_array(out[1], shape=(max(Lp[n], 1),))[:Lp[n]].copy(),
           np.ctypeslib.as_array(out[2], shape=(max(Lp[n], 1),))[:Lp[n]].copy(),
           Up, np.ctypeslib.as_array(out[4], shape=(max(Up[n], 1),))[:Up[n]].copy(),
           np.ctypeslib.as_array(out[5], shape=(max(Up[n], 1),))[:Up[n]].copy(), pinv)
    for o in out:
        lib.csx_host_free(C.cast(o, C.c_void_p))
    return res


def _device_lu(Ap, Ai, Ax, tol):
    """csx_lu_etree through the C ABI: (Lp, Li, Lx, Up, Ui, Ux, pinv) or None (singular), or 'host' when it declines."""
    import _csx
    lib = _csx.lib()
    n = len(Ap) - 1
    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    hL, hU, done = _csx.new_handle(), _csx.new_handle(), C.c_int(0)
    pinv = np.empty(n, np.int32)
    st = lib.csx_lu_etree(hA, float(tol), hL, hU, _csx.pi(pinv), done)
    _csx.free(hA)
    if st == _csx.ENOTSPD:
        return None
    _csx.check(st)
    if not done.value:
        return "host"
    out = []
    for h in (hL, hU):
        m_, n_, z_, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
        _csx.check(lib.csx_csc_info(h, m_, n_, z_, hv))
        p, i, x = np.empty(n + 1, np.int32), np.empty(max(z_.value, 1), np.int32), np.empty(max(z_.value, 1))
        _csx.check(lib.csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
        out += [p, i[:z_.value], x[:z_.value]]
        _csx.free(h)
    return tuple(out) + (pinv,)


def _same(dev, host):
    assert dev is not None and dev != "host" and host is not None
    for a, b in zip(dev, host):
        assert a.shape == b.shape and a.tobytes() == b.tobytes()


def _w_chain(nb, cs_mod):
    """SURVEY 8d's W-chain: nb blocks of the drop-tol'd west0067 pattern, values scaled per block, plus A(67 b, 67 b - 1) =
    1e-3 linking every block to the one before it: ONE connected matrix."""
    import synth
    g = golden("west0067")
    bp, bi, bx = g["C_p"].astype(np.int64), g["C_i"].astype(np.int64), g["C_x"]
    bs = 67
    n = nb * bs
    u = synth.vec(nb, 20240604, 0.0, 1.0)
    cols_i, cols_x, Ap = [], [], [0]
    for b in range(nb):
        for c in range(bs):
            ri = (bi[bp[c]:bp[c + 1]] + b * bs).tolist()
            rx = (bx[bp[c]:bp[c + 1]] * (1.0 + 1e-3 * u[b])).tolist()
            if c == bs - 1 and b + 1 < nb:                 # column 67 (b + 1) - 1 gets the row 67 (b + 1)
                ri.append((b + 1) * bs)
                rx.append(1e-3)
            cols_i += ri
            cols_x += rx
            Ap.append(len(cols_i))
    return n, np.asarray(Ap, np.int32), np.asarray(cols_i, np.int32), np.asarray(cols_x)


def _unsym_grid(g):
    """Convection-diffusion on a g x g grid, 5-point upwind stencil: unsymmetric values, symmetric pattern, one component."""
    import scipy.sparse as sp
    T = sp.diags([-1.0 - 0.7, 4.2, -1.0 + 0.7], [-1, 0, 1], shape=(g, g))
    S = sp.diags([-1.0 - 0.3, 0.0, -1.0 + 0.3], [-1, 0, 1], shape=(g, g))
    A = (sp.kron(sp.identity(g), T) + sp.kron(S, sp.identity(g))).tocsc()
    A.sort_indices()
    return g * g, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


@pytest.mark.parametrize("tol", [1.0, 0.001])
def test_w_chain_factors_on_the_device_bit_identical_to_the_host_loop(cs, tol):
    import _csx
    n, Ap, Ai, Ax = _w_chain(60, cs)
    # the links make the column elimination tree nearly a chain (one level per column of a block, block after block):
    # the planner ("lu.etree" = 1; the default since round 4 is 0 = never) hands such a matrix to the host loop ...
    assert _device_lu(Ap, Ai, Ax, tol) == "host"
    with _csx.option("lu.etree", 1):
        assert _device_lu(Ap, Ai, Ax, tol) == "host"
    # ... and when told to take it anyway the device gives the host loop's factors bit for bit
    with _csx.option("lu.etree", 2):
        dev = _device_lu(Ap, Ai, Ax, tol)
    _same(dev, _host_lu(n, Ap, Ai, Ax, tol))


@pytest.mark.parametrize("tol", [1.0, 0.1])
def test_unsymmetric_grid_in_a_dissection_order(cs, tol):
    """A grid in natural order is a band (chain tree); in the order-2 column permutation (cs_amd(2, A): nested dissection
    of A'A) the column elimination tree is bushy but its top columns reach thousands of rows: too much for one lane, the
    planner keeps both on the host.  Told to take it anyway ("lu.etree" = 2) the device gives the host loop's factors
    bit for bit; the drop-in cs_lu / cs_lusol with order 2 give the same factors and a solution with a tiny residual."""
    import _csx
    n, Ap, Ai, Ax = _unsym_grid(70)
    with _csx.option("lu.etree", 1):
        assert _device_lu(Ap, Ai, Ax, tol) == "host"                      # natural order: a chain
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    S = cs.cs_sqr(2, A, False)
    AQ = cs.cs_permute(A, None, S.q, True)
    Qp, Qi, Qx = np.asarray(AQ.p, np.int32), np.asarray(AQ.i[:AQ.p[n]], np.int32), np.asarray(AQ.x[:AQ.p[n]])
    with _csx.option("lu.etree", 1):
        assert _device_lu(Qp, Qi, Qx, tol) == "host"                      # long reaches at the top of the tree
    with _csx.option("lu.etree", 2):
        dev = _device_lu(Qp, Qi, Qx, tol)
    _same(dev, _host_lu(n, Qp, Qi, Qx, tol))
    N = cs.cs_lu(A, S, tol)
    assert N.pinv == dev[6].tolist() and N.L.p == dev[0].tolist() and N.U.x[:N.U.p[n]] == dev[5].tolist()
    b = [1.0 + i / n for i in range(n)]
    x = list(b)
    assert cs.cs_lusol(2, A, x, tol) is True
    import c_oracle as CO
    r = CO.gaxpy(n, n, Ap, Ai, Ax, np.asarray(x), -np.asarray(b))
    assert np.max(np.abs(r)) < 1e-11


def _bordered_blocks(nblocks, seed=4):
    """Components of 120 rows (more than the 96 the lane-per-block kernel takes): four unsymmetric banded sub-blocks of 28
    columns and 8 border COLUMNS that reach into all of them (the sub-blocks' columns stay inside their own rows: columns
    that share a row are a clique of A'A and would turn the tree into a chain) -- a shallow, bushy column elimination
    tree with short reaches: the shape the tree-scheduled kernel is for."""
    rng = np.random.default_rng(seed)
    bs, sub, nsub, bord = 120, 28, 4, 8
    cols_i, cols_x, Ap = [], [], [0]
    for b in range(nblocks):
        o = b * bs
        for c in range(bs):
            rows = {c}
            if c < sub * nsub:
                s0 = (c // sub) * sub
                for d in (-2, -1, 1, 2):
                    if s0 <= c + d < s0 + sub:
                        rows.add(c + d)
            else:
                rows |= set(range(sub * nsub, bs))
                rows |= {int(v) for v in rng.choice(sub * nsub, 10, replace=False)}
            rows = sorted(rows)
            vals = rng.uniform(-1.0, 1.0, len(rows))
            vals[rows.index(c)] = 4.0 + rng.uniform(0, 1)          # a strong diagonal; off-diagonals of both signs
            cols_i += [o + r for r in rows]
            cols_x += vals.tolist()
            Ap.append(len(cols_i))
    return nblocks * bs, np.asarray(Ap, np.int32), np.asarray(cols_i, np.int32), np.asarray(cols_x)


@pytest.mark.parametrize("tol", [1.0, 0.01])
def test_bordered_blocks_factor_on_the_device_by_the_planners_own_choice(cs, tol):
    """"lu.etree" = 1: shallow trees with short columns go to the device.  (Not the default: on this very shape the device
    ties with one host core at 4 000 blocks and loses below -- tools/time_lu_bordered.py, profiles/r04_ablation.md --, so
    since round 4 a connected cs_lu is a host component unless asked for.)"""
    import _csx
    n, Ap, Ai, Ax = _bordered_blocks(40)
    assert _device_lu(Ap, Ai, Ax, tol) == "host"                           # the default: never
    with _csx.option("lu.etree", 1):
        dev = _device_lu(Ap, Ai, Ax, tol)
        _same(dev, _host_lu(n, Ap, Ai, Ax, tol))
        A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
        N = cs.cs_lu(A, cs.cs_sqr(0, A, False), tol)                       # the drop-in takes the same path
        assert N.L._lazy and N.pinv == dev[6].tolist() and N.L.x[:N.L.p[n]] == dev[2].tolist()
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    b = [1.0 + i / n for i in range(n)]
    x = list(b)
    assert cs.cs_lusol(0, A, x, tol) is True
    import c_oracle as CO
    r = CO.gaxpy(n, n, Ap, Ai, Ax, np.asarray(x), -np.asarray(b))
    assert np.max(np.abs(r)) < 1e-11


def test_singular_matrix_is_reported_like_the_host_loop(cs):
    n, Ap, Ai, Ax = _unsym_grid(60)
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    q = cs.cs_amd(2, A)
    AQ = cs.cs_permute(A, None, q, True)
    Qp, Qi, Qx = np.asarray(AQ.p, np.int32), np.asarray(AQ.i[:AQ.p[n]], np.int32), np.asarray(AQ.x[:AQ.p[n]]).copy()
    col = 1234
    Qx[Qp[col]:Qp[col + 1]] = 0.0                                          # a column of explicit zeros
    assert _host_lu(n, Qp, Qi, Qx, 1.0) is None
    import _csx
    with _csx.option("lu.etree", 2):
        assert _device_lu(Qp, Qi, Qx, 1.0) is None
