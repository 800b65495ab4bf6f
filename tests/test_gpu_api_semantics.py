"""Drop-in semantics of the product module on degenerate and unusual inputs, checked against the
oracle (which is pinned to the reference): empty matrices, zero-length vectors, numpy-backed cs
fields, aliasing, device-resident objects, lazy materialisation."""
import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, unpack
from test_gpu_parity import cs  # noqa: F401

pytestmark = pytest.mark.gpu


def mk(mod, m, n, p, i, x):
    A = mod.cs_spalloc(m, n, len(i), x is not None, False)
    A.p, A.i = list(p), list(i) if len(i) else [0]
    A.x = None if x is None else (list(x) if len(x) else [0.0])
    return A


def same(C, R):
    assert (C.m, C.n, C.nz, C.nzmax) == (R.m, R.n, R.nz, R.nzmax)
    assert list(C.p) == list(R.p) and list(C.i) == list(R.i)
    assert (C.x is None) == (R.x is None)
    if R.x is not None:
        assert list(C.x) == list(R.x)


CASES = [
    (3, 4, [0, 0, 0, 0, 0], [], []),                 # no entries at all
    (1, 1, [0, 1], [0], [2.5]),                      # 1 x 1
    (5, 3, [0, 0, 2, 2], [4, 0], [1.5, -2.0]),       # empty first and last column, unsorted rows
    (2, 6, [0, 1, 1, 3, 3, 3, 5], [1, 0, 1, 0, 0], [1.0, 2.0, 3.0, 4.0, 5.0]),  # duplicate (0,5)
]


@pytest.mark.parametrize("m,n,p,i,x", CASES)
def test_degenerate_matrices(cs, m, n, p, i, x):
    A, Ao = mk(cs, m, n, p, i, x), mk(O, m, n, p, i, x)
    same(cs.cs_transpose(A, True), O.cs_transpose(Ao, True))
    same(cs.cs_transpose(A, False), O.cs_transpose(Ao, False))
    AT, ATo = cs.cs_transpose(A, True), O.cs_transpose(Ao, True)
    same(cs.cs_multiply(A, AT), O.cs_multiply(Ao, ATo))
    same(cs.cs_multiply(AT, A), O.cs_multiply(ATo, Ao))
    xv = [0.5 * (k + 1) for k in range(n)]
    y, yo = [1.0] * m, [1.0] * m
    assert cs.cs_gaxpy(A, xv, y) is True and O.cs_gaxpy(Ao, xv, yo) is True
    assert y == yo


def test_zero_dimension_matrices(cs):
    for m, n in ((0, 0), (0, 3), (3, 0)):
        A, Ao = mk(cs, m, n, [0] * (n + 1), [], []), mk(O, m, n, [0] * (n + 1), [], [])
        same(cs.cs_transpose(A, True), O.cs_transpose(Ao, True))
        y = [7.0] * m
        assert cs.cs_gaxpy(A, [1.0] * n, y) is True and y == [7.0] * m
        same(cs.cs_multiply(A, cs.cs_transpose(A, True)), O.cs_multiply(Ao, O.cs_transpose(Ao, True)))
    L = mk(cs, 0, 0, [0], [], [])
    assert cs.cs_lsolve(L, []) is True and cs.cs_usolve(L, []) is True


def test_numpy_backed_fields_and_aliasing(cs):
    g = golden("west0067")
    A = unpack(cs, g, "A")
    A.p, A.i, A.x = np.asarray(A.p), np.asarray(A.i), np.asarray(A.x)     # arrays instead of lists
    x = g["gaxpy_x"]
    y = g["gaxpy_y0"].copy()
    assert cs.cs_gaxpy(A, x, y) is True
    assert y.tobytes() == g["gaxpy_y"].tobytes()
    # x and y may be the same object for a square matrix: the reference reads x[j] while it writes y;
    # on the device x is uploaded before y is touched, so only the non-aliased result is defined here
    z = g["gaxpy_y0"].tolist()
    zz = z
    assert cs.cs_gaxpy(A, g["gaxpy_x"].tolist(), z) and zz is z
    AT = cs.cs_transpose(A, True)
    assert AT.p == g["AT_p"].tolist() and isinstance(AT.p, list) and isinstance(AT.x, list)


def test_pinned_and_lazy_objects(cs):
    g = golden("bcsstk01")
    A = cs.cs_pin(unpack(cs, g, "A"))
    AT = cs.cs_transpose(A, True)                    # stays on the device: nothing copied yet
    assert AT._lazy and (AT.m, AT.n, AT.nzmax) == (48, 48, 224)
    C = cs.cs_multiply(A, AT)                        # device -> device
    assert C._lazy and C.nzmax == 764
    assert C.p[48] == 764 and not C._lazy            # first read materialises lists
    assert C.i[:764] == g["AAT_i"].tolist() and len(C.x) == 764
    # reassigning a field of a pinned matrix drops the stale device copy
    x = g["gaxpy_x"].tolist()
    y1 = [0.0] * 48
    cs.cs_gaxpy(A, x, y1)
    A.x = [2.0 * v for v in A.x]
    assert A._dev is None
    y2 = [0.0] * 48
    cs.cs_gaxpy(A, x, y2)
    assert y2 == [2.0 * v for v in y1]
    # in-place edits need cs_invalidate (documented)
    B = cs.cs_pin(unpack(cs, g, "A"))
    B.x[0] *= 3.0
    cs.cs_invalidate(B)
    y3, yo = [0.0] * 48, [0.0] * 48
    cs.cs_gaxpy(B, x, y3)
    Bo = unpack(O, g, "A")
    Bo.x[0] *= 3.0
    O.cs_gaxpy(Bo, x, yo)
    assert y3 == yo


def test_dvec_block_roundtrip_and_shapes(cs):
    B = np.arange(12, dtype=np.float64).reshape(4, 3)
    d = cs.dvec(B)
    assert (d.n, d.k, len(d)) == (4, 3, 4) and d.numpy().tolist() == B.tolist()
    c = d.copy()
    d.fill(1.5)
    assert c.numpy().tolist() == B.tolist() and set(d.numpy().ravel().tolist()) == {1.5}
    d.assign(2 * B)
    assert d.tolist() == (2 * B).tolist()
    z = cs.dvec(5)
    assert z.tolist() == [0.0] * 5 and isinstance(z.device_ptr(), int)
    with pytest.raises(IndexError):
        cs.cs_lsolve(cs.cs_pin(unpack(cs, golden("t1"), "A")), cs.dvec(3))   # x shorter than n


def test_one_shot_host_gaxpy_through_the_c_abi(cs):
    import _csx
    g = golden("bcsstk16")
    A = unpack(cs, g, "A")
    nnz = A.p[A.n]
    p, i, x = _csx.i32(A.p), _csx.i32(A.i[:nnz]), _csx.f64(A.x[:nnz])
    xv, yv = _csx.f64(g["gaxpy_x"]), _csx.f64(g["gaxpy_y0"]).copy()
    _csx.check(_csx.lib().csx_gaxpy_host(A.m, A.n, _csx.pi(p), _csx.pi(i), _csx.pd(x), _csx.pd(xv), _csx.pd(yv)))
    assert yv.tobytes() == g["gaxpy_y"].tobytes()          # the unmodified reference's y, bit for bit
    assert _csx.lib().csx_gaxpy_host(A.m, A.n, _csx.pi(p), _csx.pi(i), _csx.pd(x), None, _csx.pd(yv)) == _csx.EINVAL


def test_full_cache_keeps_the_block_just_released_and_drops_the_oldest(cs):
    """The device-memory cache under pressure: with room for one 40 MB block, releasing A then B must leave B cached (A
    goes back to the driver), and the next request of that size must be served from the cache.  (Freeing the newcomer
    instead cost cs_multiply a hipFree + hipMalloc of its 12 GB work arrays per call in a full cache.)"""
    import _csx
    lib = _csx.lib()
    C = _csx.C

    def info():
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _csx.check(lib.csx_mem_info(a, b, c), "mem_info")
        return a.value, b.value

    _csx.check(lib.csx_mem_trim(), "trim")
    n = 5 * 1000 * 1000                                    # 40 MB of doubles
    with _csx.option("pool.limit_mb", 64):
        hA, hB = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(n, hA), "alloc")
        _csx.check(lib.csx_vec_alloc(n, hB), "alloc")
        cached0, live0 = info()
        _csx.free(hA)
        cached1, _ = info()
        assert cached1 - cached0 >= 8 * n                  # A is cached
        _csx.free(hB)
        cached2, _ = info()
        assert 8 * n <= cached2 - cached0 < 16 * n          # room for one: B stayed, A went back to the driver
        hC = _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(n, hC), "alloc")
        cached3, _ = info()
        assert cached3 - cached0 < 8 * n                   # served from the cache
        _csx.free(hC)
    _csx.check(lib.csx_mem_trim(), "trim")


def test_option_blocks_nest_and_restore_what_was_in_force(cs):
    import _csx
    lib = _csx.lib()
    v = _csx.C.c_int(-7)

    def get(name):
        _csx.check(lib.csx_get_option(name.encode(), v))
        return v.value

    assert get("chol.wband") == 1 and get("tri.levels_where") == 0 and get("chol.wband_nb") == 16
    _csx.check(lib.csx_set_option(b"chol.wband", 2))            # a process-wide non-default setting ...
    with _csx.option("chol.wband", 0):
        assert get("chol.wband") == 0
        with _csx.option("chol.wband", 1):
            assert get("chol.wband") == 1
        assert get("chol.wband") == 0                           # ... inner block restores the outer block's value
    assert get("chol.wband") == 2                               # ... and the outer one the process-wide one
    _csx.check(lib.csx_set_option(b"chol.wband", 1))
    with _csx.option("tri.levels_where", 5):                    # normalised by the library
        assert get("tri.levels_where") == 0
    assert lib.csx_get_option(b"no.such.option", v) == _csx.EINVAL
    assert lib.csx_set_option(b"no.such.option", 1) == _csx.EINVAL
