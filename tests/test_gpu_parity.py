"""Parity of the HIP path (through the C ABI, via the drop-in module) against
the pinned CPU oracle and the committed golden vectors.  Needs an MI355X."""
import numpy as np
import pytest

import c_oracle as CO
import csparse_oracle as O
import synth
from conftest import golden, unpack, same_csc

pytestmark = pytest.mark.gpu

SMALL = ["t1", "bcsstk01", "west0067", "ash219", "fs_183_1", "ibm32a", "ibm32b", "lp_afiro"]
ALL = SMALL + ["bcsstk16", "mbeacxc"]
RTOL = 1e-10  # BASELINE.json north_star: x[] within 1e-10 relative


@pytest.fixture(scope="module")
def cs():
    import csparse
    import _csx
    _csx.init()
    return csparse


def rel_err(got, ref, scale=None, nterms=1024):
    """max |got-ref| / (|ref| + (nterms*eps/RTOL) * scale).

    `scale` = sum of |terms| behind each result.  Summing k terms in another order than
    the reference moves a result by at most ~k*eps*sum|terms|, whatever the result's own
    size (it may be 0 after cancellation), so that much is granted on top of the 1e-10
    relative budget of BASELINE.json; `nterms` bounds k."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    den = np.abs(ref)
    if scale is not None:
        den = den + (nterms * np.finfo(np.float64).eps / RTOL) * np.asarray(scale)
    den = np.where(den == 0, 1.0, den)
    return float(np.max(np.abs(got - ref) / den)) if got.size else 0.0


def abs_terms(A, x):
    """sum_j |A(i,j) x_j| per row: the scale that bounds reordering error"""
    nnz = A.p[A.n]
    Ap = np.asarray(A.p, dtype=np.int32)
    return CO.gaxpy(A.m, A.n, Ap, np.asarray(A.i[:nnz], dtype=np.int32), np.abs(np.asarray(A.x[:nnz])),
                    np.abs(np.asarray(x, dtype=np.float64)), np.zeros(A.m))


# ---------------------------------------------------------------- gaxpy ----

@pytest.mark.parametrize("name", ALL)
def test_gaxpy_list_call_is_bit_exact(cs, name):
    g = golden(name)
    A = unpack(cs, g, "A")
    x = g["gaxpy_x"].tolist()
    y = g["gaxpy_y0"].tolist()
    alias = y
    assert cs.cs_gaxpy(A, x, y) is True
    assert alias is y and np.asarray(y).tobytes() == g["gaxpy_y"].tobytes()
    assert x == g["gaxpy_x"].tolist()


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("mode", ["WAVE", "ATOMIC", "TILED"])
def test_gaxpy_fast_modes_within_tolerance(cs, name, mode):
    g = golden(name)
    A = unpack(cs, g, "A")
    cs.cs_pin(A)
    dx, dy = cs.dvec(g["gaxpy_x"]), cs.dvec(g["gaxpy_y0"])
    assert cs.cs_gaxpy(A, dx, dy, getattr(cs, "GAXPY_" + mode)) is True
    scale = abs_terms(A, g["gaxpy_x"]) + np.abs(g["gaxpy_y0"])
    assert rel_err(dy.numpy(), g["gaxpy_y"], scale) < RTOL


def test_gaxpy_edge_cases(cs):
    g = golden("synthetic_20240601")
    for c, (m, n, nnz) in enumerate(g["cases"]):
        pre = "c%d_" % c
        A = unpack(cs, g, pre + "A")
        y = g[pre + "y0"].tolist()
        assert cs.cs_gaxpy(A, g[pre + "x"].tolist(), y)
        assert np.asarray(y).tobytes() == g[pre + "y"].tobytes()
        for mode in (cs.GAXPY_WAVE, cs.GAXPY_ATOMIC, cs.GAXPY_TILED):
            B = cs.cs_pin(unpack(cs, g, pre + "A"))
            dy = cs.dvec(g[pre + "y0"])
            assert cs.cs_gaxpy(B, cs.dvec(g[pre + "x"]), dy, mode)
            scale = abs_terms(B, g[pre + "x"]) + np.abs(g[pre + "y0"])
            assert rel_err(dy.numpy(), g[pre + "y"], scale) < RTOL
    # numpy y is updated in place too; longer-than-needed vectors are fine (len >= n)
    A = unpack(cs, g, "c0_A")
    y = np.concatenate([g["c0_y0"], [7.0]])
    assert cs.cs_gaxpy(A, np.concatenate([g["c0_x"], [9.0]]), y)
    assert y[:-1].tobytes() == g["c0_y"].tobytes() and y[-1] == 7.0
    with pytest.raises(IndexError):
        cs.cs_gaxpy(A, [1.0], [0.0] * A.m)
    bad = unpack(cs, g, "c0_A")
    bad.i[0] = bad.m  # out of range row
    with pytest.raises(IndexError):
        cs.cs_gaxpy(bad, [0.0] * bad.n, [0.0] * bad.m)


# ------------------------------------------------------------ transpose ----

@pytest.mark.parametrize("name", ALL)
def test_transpose_bit_exact(cs, name):
    g = golden(name)
    A = unpack(cs, g, "A")
    AT = cs.cs_transpose(A, True)
    same_csc(AT, g, "AT")
    P = cs.cs_transpose(A, False)
    assert P.x is None and P.i[:P.p[P.n]] == g["AT_i"].tolist() and P.p == g["AT_p"].tolist()
    # transpose twice sorts the columns and keeps the multiset of entries
    ATT = cs.cs_transpose(AT, True)
    Oa = unpack(O, g, "A")
    ref = O.cs_transpose(O.cs_transpose(Oa, True), True)
    assert ATT.p == ref.p and ATT.i == ref.i and ATT.x == ref.x


def test_transpose_edge_cases(cs):
    g = golden("synthetic_20240601")
    for c, (m, n, nnz) in enumerate(g["cases"]):
        pre = "c%d_" % c
        A = unpack(cs, g, pre + "A")
        same_csc(cs.cs_transpose(A, True), g, pre + "AT")
        A.x = None
        T = cs.cs_transpose(A, True)  # values requested but absent -> pattern (csparse.py:2302)
        assert T.x is None and T.p == g[pre + "AT_p"].tolist()


# ------------------------------------------- generators and medium sizes ----

def test_generators_match_host_twins(cs):
    import _csx
    h = _csx.new_handle()
    _csx.check(_csx.lib().csx_gen_grand(1000, 16, 77, h))
    p, i, x = np.empty(1001, np.int32), np.empty(16000, np.int32), np.empty(16000)
    _csx.check(_csx.lib().csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
    rp, ri, rx = synth.grand(1000, 16, 77)
    assert p.tolist() == rp.tolist() and i.tolist() == ri.tolist() and x.tobytes() == rx.tobytes()
    _csx.free(h)
    # SURVEY 8d's draw, the one bench.py's headline is timed on: per_col DISTINCT uniform rows per column, ascending
    for n, pc, seed in ((30000, 64, 5), (1000, 32, 9), (200, 64, 1), (5000, 7, 3), (128, 64, 20240602)):
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_gen_grand_uniform(n, pc, seed, h))
        p, i, x = np.empty(n + 1, np.int32), np.empty(n * pc, np.int32), np.empty(n * pc)
        _csx.check(_csx.lib().csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
        rp, ri, rx = synth.grand_uniform(n, pc, seed)
        assert (p == rp).all() and (i == ri).all() and x.tobytes() == rx.tobytes(), (n, pc, seed)
        rows = i.reshape(n, pc)
        assert (np.diff(rows, axis=1) > 0).all() and rows.min() >= 0 and rows.max() < n   # distinct, ascending, in range
        _csx.free(h)
    for bs in (8, 64):
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_gen_gspd(5, bs, 99, h))
        n = 5 * bs
        p, i, x = np.empty(n + 1, np.int32), np.empty(n * bs, np.int32), np.empty(n * bs)
        _csx.check(_csx.lib().csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
        rp, ri, rx = synth.gspd(5, bs, 99)
        assert p.tolist() == rp.tolist() and i.tolist() == ri.tolist() and x.tobytes() == rx.tobytes()
        _csx.free(h)
    h = _csx.new_handle()
    _csx.check(_csx.lib().csx_gen_vec(1234, 5, 0.5, 1.5, h))
    assert cs.dvec(1234, 1, _handle=h).numpy().tobytes() == synth.vec(1234, 5, 0.5, 1.5).tobytes()
    h = _csx.new_handle()
    _csx.check(_csx.lib().csx_gen_rhs(100, 3, 4, h))
    assert cs.dvec(100, 3, _handle=h).numpy().tobytes() == synth.rhs(100, 3, 4).tobytes()


def _host_cs(cs, m, n, Ap, Ai, Ax):
    A = cs.cs_spalloc(m, n, len(Ai), True, False)
    A.p, A.i, A.x = Ap.tolist(), Ai.tolist(), Ax.tolist()
    return A


@pytest.mark.parametrize("n,per_col", [(20000, 64), (50000, 7), (4099, 130)])
def test_grand_medium_against_c_oracle(cs, n, per_col):
    Ap, Ai, Ax = synth.grand(n, per_col, 20240602)
    x = synth.vec(n, 11, 0.5, 1.5)
    y0 = synth.vec(n, 12, -1.0, 1.0)
    ref = CO.gaxpy(n, n, Ap, Ai, Ax, x, y0)
    A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
    dy = cs.dvec(y0)
    assert cs.cs_gaxpy(A, cs.dvec(x), dy, cs.GAXPY_EXACT)
    assert dy.numpy().tobytes() == ref.tobytes()
    scale = CO.gaxpy(n, n, Ap, Ai, np.abs(Ax), np.abs(x), np.abs(y0))
    for mode in (cs.GAXPY_WAVE, cs.GAXPY_ATOMIC, cs.GAXPY_TILED, cs.GAXPY_AUTO):
        dy = cs.dvec(y0)
        assert cs.cs_gaxpy(A, cs.dvec(x), dy, mode)
        assert rel_err(dy.numpy(), ref, scale) < RTOL, mode
    Tp, Ti, Tx = CO.transpose(n, n, Ap, Ai, Ax)
    AT = cs.cs_transpose(A, True)
    assert AT.p == Tp.tolist() and AT.i == Ti.tolist()
    assert np.asarray(AT.x).tobytes() == Tx.tobytes()


def test_transpose_wide_keys_three_radix_passes(cs):
    """m > 65536 rows needs three 8-bit passes; duplicates must keep source order."""
    rng = np.random.default_rng(5)
    m, n, nnz = 200003, 300, 60000
    cols = np.sort(rng.integers(0, n, nnz))
    rows = rng.integers(0, m, nnz)
    rows[::7] = rows[1::7][: len(rows[::7])]  # plant duplicates
    vals = rng.standard_normal(nnz)
    Ap = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int32)
    Tp, Ti, Tx = CO.transpose(m, n, Ap, rows.astype(np.int32), vals)
    A = _host_cs(cs, m, n, Ap, rows.astype(np.int32), vals)
    AT = cs.cs_transpose(A, True)
    assert AT.m == n and AT.n == m
    assert AT.p == Tp.tolist() and AT.i == Ti.tolist() and np.asarray(AT.x).tobytes() == Tx.tobytes()
    # (with values and 17 .. 24 key bits the keys travel as 16 bits after the first pass, the spent digit in the top bits
    # of the column word; "sort.short_keys" = 0 keeps them 32 bits wide: the same result)
    import _csx
    with _csx.option("sort.short_keys", 0):
        AT0 = cs.cs_transpose(A, True)
    assert AT0.p == Tp.tolist() and AT0.i == Ti.tolist() and np.asarray(AT0.x).tobytes() == Tx.tobytes()


def _transpose_abi(A_np, values=True):
    """cs_transpose through the C ABI on numpy arrays (no Python lists: for cases with millions of rows)."""
    import _csx
    m, n, Ap, Ai, Ax = A_np
    lib = _csx.lib()
    hA, hT = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_csc_upload(m, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    _csx.check(lib.csx_transpose(hA, 1 if values else 0, hT))
    nnz = int(Ap[-1])
    Tp, Ti, Tx = np.empty(m + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    _csx.check(lib.csx_csc_download(hT, _csx.pi(Tp), _csx.pi(Ti), _csx.pd(Tx) if values else None))
    _csx.free(hA)
    _csx.free(hT)
    return Tp, Ti, Tx


@pytest.mark.parametrize("m,n,nnz,what", [
    (41, 30011, 3000, "thousands of empty columns start inside one tile: columns searched in the pointer array"),
    (5000, 9000, 30000, "about 1 200 column starts per tile: starts listed in LDS and searched"),
    (70001, 400, 50000, "about 30 starts per tile: columns painted from the tile descriptor; three-block pointer scan"),
    (17000001, 7, 9000, "25-bit row keys: four passes"),
    (300, 3, 20000, "three long columns: most tiles lie inside one column"),
])
def test_transpose_column_expansion_paths(cs, m, n, nnz, what):
    """The first radix pass derives each entry's column from its position (csparse.py:2308-2314 walks the columns);
    every way the kernel does that, with duplicates and unsorted rows, against the C oracle, bit for bit."""
    rng = np.random.default_rng(m + n)
    cols = np.sort(rng.integers(0, n, nnz))
    if n > 10:
        cols[cols < n // 10] = n // 10                      # empty columns in front ...
        cols[cols > n - n // 20] = n - n // 20              # ... and at the end
    rows = rng.integers(0, m, nnz).astype(np.int32)
    rows[::5] = rows[1::5][: len(rows[::5])]                # duplicates, source order must survive
    vals = rng.standard_normal(nnz)
    Ap = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int32)
    Rp, Ri, Rx = CO.transpose(m, n, Ap, rows, vals)
    Tp, Ti, Tx = _transpose_abi((m, n, Ap, rows, vals))
    assert np.array_equal(Tp, Rp) and np.array_equal(Ti, Ri) and Tx.tobytes() == Rx.tobytes(), what
    Tp, Ti, _ = _transpose_abi((m, n, Ap, rows, vals), values=False)
    assert np.array_equal(Tp, Rp) and np.array_equal(Ti, Ri), what


@pytest.mark.parametrize("n,per_col,expect", [(65536, 64, 3), (60000, 64, 4), (30000, 3, 4)])
def test_tiled_plan_key_formats(cs, n, per_col, expect):
    """The LDS-tiled cs_gaxpy plan stores 3-byte keys (row + 9-bit column offset inside a run of 64 column-sorted
    entries) when every run is narrower than 512 columns, 4-byte keys otherwise.  Dense-enough random matrices take
    the first (65 536 rows = 256 equal row blocks), sparse ones the second (a run of 64 entries of a tile spans
    thousands of columns; at 60 000 rows the last row block has 75 rows instead of 235); both must give the
    oracle's y (rounding-equal: the tiled plan accumulates with LDS atomics)."""
    import _csx
    Ap, Ai, Ax = synth.grand(n, per_col, 20240615)
    x = synth.vec(n, 11, 0.5, 1.5)
    y0 = synth.vec(n, 12, -1.0, 1.0)
    ref = CO.gaxpy(n, n, Ap, Ai, Ax, x, y0)
    scale = CO.gaxpy(n, n, Ap, Ai, np.abs(Ax), np.abs(x), np.abs(y0))
    lib = _csx.lib()
    out = {}
    for keys24 in (1, 0):
        _csx.check(lib.csx_set_option(b"gaxpy.keys24", keys24))
        try:
            A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
            dy = cs.dvec(y0)
            assert cs.cs_gaxpy(A, cs.dvec(x), dy, cs.GAXPY_TILED) is True
            kb = _csx.C.c_int(0)
            _csx.check(lib.csx_gaxpy_plan_info(A._dev.handle, None, None, kb))
            assert kb.value == (expect if keys24 else 4)
            out[keys24] = dy.numpy()
            assert rel_err(out[keys24], ref, scale) < RTOL
        finally:
            _csx.check(lib.csx_set_option(b"gaxpy.keys24", 1))
    assert rel_err(out[1], out[0], scale) < RTOL     # LDS atomics: the two runs agree to rounding, not bit for bit
