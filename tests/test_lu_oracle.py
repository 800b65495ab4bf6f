"""cs_lu (csparse.py:1370-1451) -- the product's host loop (csx_lu_host, C++) against the oracle's restatement of the
reference loop (oracle/csparse_oracle.py cs_lu, the walk bounded to xi[top..n-1] as the loop's own comment requires;
SURVEY D7): L, U and pinv bit for bit on the reference's matrices and a seeded unsymmetric grid.  The device kernels are
compared with csx_lu_host (tests/test_gpu_lu_blocks.py, tests/test_gpu_lu_etree.py) and, below, with the oracle directly.
The oracle's cs_lu itself is pinned by the unmodified reference's cs_lusol answers (tests/test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack


def host_lu(n, Ap, Ai, Ax, tol):
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


def oracle_lu(n, Ap, Ai, Ax, tol):
    """(Lp, Li, Lx, Up, Ui, Ux, pinv) of the oracle's cs_lu at order 0."""
    A = O.cs_spalloc(n, n, max(len(Ai), 1), True, False)
    A.p, A.i, A.x = [int(v) for v in Ap], [int(v) for v in Ai], [float(v) for v in Ax]
    S = O.cs_sqr(0, A, False)
    N = O.cs_lu(A, S, tol)
    if N is None:
        return None
    lnz, unz = N.L.p[n], N.U.p[n]
    return (np.asarray(N.L.p, np.int32), np.asarray(N.L.i[:lnz], np.int32), np.asarray(N.L.x[:lnz], np.float64),
            np.asarray(N.U.p, np.int32), np.asarray(N.U.i[:unz], np.int32), np.asarray(N.U.x[:unz], np.float64),
            np.asarray(N.pinv, np.int32))


def unsym_grid(g, seed):
    """Convection-diffusion on a g x g grid with seeded coefficients: unsymmetric values, weak diagonals here and there (so
    the threshold pivoting has choices to make), one connected component."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = g * g
    T = sp.diags([-1.0 - 0.7, 4.2, -1.0 + 0.7], [-1, 0, 1], shape=(g, g))
    S = sp.diags([-1.0 - 0.3, 0.0, -1.0 + 0.3], [-1, 0, 1], shape=(g, g))
    A = (sp.kron(sp.identity(g), T) + sp.kron(S, sp.identity(g))).tocsc()
    A.sort_indices()
    x = A.data * rng.uniform(0.5, 1.5, A.nnz)
    cols = np.repeat(np.arange(n), np.diff(A.indptr))
    weak = (A.indices == cols) & (rng.uniform(size=A.nnz) < 0.2)
    x[weak] *= 0.05
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), x


def same(got, ref):
    assert got is not None and ref is not None
    for a, b, what in zip(got, ref, ("L.p", "L.i", "L.x", "U.p", "U.i", "U.x", "pinv")):
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), what


CASES = [("t1", 1.0), ("t1", 0.001), ("west0067", 1.0), ("west0067", 0.001), ("fs_183_1", 1.0), ("fs_183_1", 0.001),
         ("bcsstk01", 0.001), ("grid12", 1.0), ("grid12", 0.1)]


def matrix(name):
    if name.startswith("grid"):
        return unsym_grid(int(name[4:]), 20240608)
    g = golden(name)
    C_ = unpack(O, g, "C")
    n = C_.n
    nnz = C_.p[n]
    return n, np.asarray(C_.p, np.int32), np.asarray(C_.i[:nnz], np.int32), np.asarray(C_.x[:nnz], np.float64)


@pytest.mark.parametrize("name,tol", CASES)
def test_host_lu_has_the_oracles_bits(name, tol):
    n, Ap, Ai, Ax = matrix(name)
    same(host_lu(n, Ap, Ai, Ax, tol), oracle_lu(n, Ap, Ai, Ax, tol))


@pytest.mark.gpu
@pytest.mark.parametrize("name,tol", CASES)
def test_device_lu_by_the_column_etree_has_the_oracles_bits(name, tol):
    """csx_lu_etree forced ("lu.etree" = 2; the planner would keep these small or deep matrices on the host)."""
    import _csx
    from test_gpu_lu_etree import _device_lu
    _csx.init(0)
    n, Ap, Ai, Ax = matrix(name)
    with _csx.option("lu.etree", 2):
        dev = _device_lu(Ap, Ai, Ax, tol)
    assert dev != "host"
    same(dev, oracle_lu(n, Ap, Ai, Ax, tol))
