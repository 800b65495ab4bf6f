"""Parity of the device SpGEMM (cs_multiply) against the oracle and golden vectors:
p[] and i[] bit-exact (first-touch column order), x[] within 1e-10."""
import hashlib

import numpy as np
import pytest

import c_oracle as CO
import synth
from conftest import golden, unpack, same_csc
from test_gpu_parity import ALL, SMALL, RTOL, _host_cs, cs, rel_err  # noqa: F401

pytestmark = pytest.mark.gpu


def sha(a, dt):
    return hashlib.sha256(np.asarray(a, dtype=dt).tobytes()).hexdigest()


def check_product(C, Cp, Ci, Cx, Sx):
    """C (product module) against oracle arrays; Sx = sum of |products| per entry (error scale)."""
    nnz = int(Cp[-1])
    assert C.p == Cp.tolist()
    assert C.i[:nnz] == Ci.tolist()
    assert C.nzmax == nnz and len(C.i) == nnz
    if Cx is None:
        assert C.x is None
        return
    assert rel_err(C.x, Cx, Sx) < RTOL


@pytest.mark.parametrize("name", ALL)
def test_multiply_a_at_reference_matrices(cs, name, meta):
    g = golden(name)
    A = unpack(cs, g, "A")
    AT = unpack(cs, g, "AT")
    C = cs.cs_multiply(A, AT)
    mm = meta[name]["AAT"]
    nnz = C.p[C.n]
    assert (C.m, C.n, nnz, C.nzmax, len(C.i)) == (mm["m"], mm["n"], mm["nnz"], mm["nzmax"], mm["leni"])
    assert sha(C.p, np.int64) == mm["sha_p"] and sha(C.i[:nnz], np.int64) == mm["sha_i"]
    ap, ai, ax = np.asarray(A.p, np.int32), np.asarray(A.i[:A.p[A.n]], np.int32), np.asarray(A.x[:A.p[A.n]])
    tp, ti, tx = np.asarray(AT.p, np.int32), np.asarray(AT.i[:AT.p[AT.n]], np.int32), np.asarray(AT.x[:AT.p[AT.n]])
    Cp, Ci, Cx = CO.multiply(A.m, A.n, AT.n, ap, ai, ax, tp, ti, tx)
    _, _, Sx = CO.multiply(A.m, A.n, AT.n, ap, ai, np.abs(ax), tp, ti, np.abs(tx))
    check_product(C, Cp, Ci, Cx, Sx)
    if name in SMALL:
        same_csc(C, g, "AAT", exact_x=False)
        P = cs.cs_multiply(cs.cs_transpose(A, False), A)  # pattern only
        same_csc(P, g, "ATA_pat")


def test_multiply_edge_cases(cs):
    g = golden("synthetic_20240601")
    for c, (m, n, nnz) in enumerate(g["cases"]):
        pre = "c%d_" % c
        A, AT = unpack(cs, g, pre + "A"), unpack(cs, g, pre + "AT")
        same_csc(cs.cs_multiply(A, AT), g, pre + "AAT", exact_x=False)
        same_csc(cs.cs_multiply(AT, A), g, pre + "ATA", exact_x=False)
        Ap = unpack(cs, g, pre + "A")
        Ap.x = None
        same_csc(cs.cs_multiply(Ap, AT), g, pre + "AAT_pat")
        if m != n:
            assert cs.cs_multiply(A, A) is None


@pytest.mark.parametrize("n,per_col,draw", [(20000, 32, "stratified"), (9000, 5, "stratified"), (70000, 12, "stratified"),
                                              (60000, 32, "uniform"), (9000, 32, "uniform")])
def test_multiply_hash_paths_against_c_oracle(cs, n, per_col, draw):
    """m > 8192 rows: the LDS hash accumulators (and, for the dense column below, the global one).  "uniform" is the draw
    of BASELINE config 4's matrix S as bench_configs.py generates it (csx_gen_grand_uniform: ragged rows, colliding
    products); at n = 9 000 a column of A A' collects ~1 000 products on ~950 rows: collisions in every column."""
    Ap, Ai, Ax = (synth.grand if draw == "stratified" else synth.grand_uniform)(n, per_col, 31)
    Tp, Ti, Tx = CO.transpose(n, n, Ap, Ai, Ax)
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    AT = _host_cs(cs, n, n, Tp, Ti, Tx)
    Cp, Ci, Cx = CO.multiply(n, n, n, Ap, Ai, Ax, Tp, Ti, Tx)
    _, _, Sx = CO.multiply(n, n, n, Ap, Ai, np.abs(Ax), Tp, Ti, np.abs(Tx))
    check_product(cs.cs_multiply(A, AT), Cp, Ci, Cx, Sx)


def test_multiply_one_huge_column_uses_global_accumulator(cs):
    n = 30000
    Ap, Ai, Ax = synth.grand(n, 8, 77)
    # B: column 0 selects 3000 columns of A (24000 products > any LDS table), others are sparse
    rng = np.random.default_rng(1)
    sel = np.sort(rng.choice(n, 3000, replace=False)).astype(np.int32)
    rest = rng.integers(0, n, size=(n - 1) * 2).astype(np.int32)
    Bi = np.concatenate([sel, rest])
    Bp = np.concatenate([[0], 3000 + 2 * np.arange(n)]).astype(np.int32)
    Bx = rng.uniform(0.5, 1.5, len(Bi))
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    B = _host_cs(cs, n, n, Bp, Bi, Bx)
    Cp, Ci, Cx = CO.multiply(n, n, n, Ap, Ai, Ax, Bp, Bi, Bx)
    _, _, Sx = CO.multiply(n, n, n, Ap, Ai, np.abs(Ax), Bp, Bi, np.abs(Bx))
    check_product(cs.cs_multiply(A, B), Cp, Ci, Cx, Sx)
    # long B column (> one staging segment of 1024 entries) through the LDS-dense path
    m2 = 500
    Ap2, Ai2, Ax2 = synth.grand(m2, 5, 5)
    Bi2 = rng.integers(0, m2, size=2500).astype(np.int32)
    Bp2 = np.asarray([0, 2500], dtype=np.int32)
    Bx2 = rng.uniform(-1, 1, 2500)
    C2 = cs.cs_multiply(_host_cs(cs, m2, m2, Ap2, Ai2, Ax2), _host_cs(cs, m2, 1, Bp2, Bi2, Bx2))
    Cp2, Ci2, Cx2 = CO.multiply(m2, m2, 1, Ap2, Ai2, Ax2, Bp2, Bi2, Bx2)
    _, _, Sx2 = CO.multiply(m2, m2, 1, Ap2, Ai2, np.abs(Ax2), Bp2, Bi2, np.abs(Bx2))
    check_product(C2, Cp2, Ci2, Cx2, Sx2)


def _ragged(rng, m, n, lens):
    """CSC with the given column lengths; rows unsorted, duplicates within a column allowed."""
    p = np.zeros(n + 1, np.int32)
    p[1:] = np.cumsum(lens)
    i = rng.integers(0, m, size=int(p[-1])).astype(np.int32)
    x = rng.uniform(-1.0, 1.0, size=int(p[-1]))
    return p, i, x


@pytest.mark.parametrize("two_pass", [False, True])
def test_multiply_ragged_columns_hash_bins(cs, two_pass):
    """m > 8192 with ragged shapes: A columns longer than one 32-lane chunk (and empty ones), B columns
    longer than one staged segment, duplicate rows inside a column, all four hash table sizes.  Both the
    one-pass kernel (bitmap ranking) and the two-pass kernel must reproduce the reference's column order."""
    import _csx
    _csx.check(_csx.lib().csx_set_option(b"spgemm.one_pass", 0 if two_pass else 1))
    try:
        _ragged_body(cs)
    finally:
        _csx.check(_csx.lib().csx_set_option(b"spgemm.one_pass", 1))


def _ragged_body(cs):
    rng = np.random.default_rng(20240611)
    m, k, n = 12000, 3000, 400
    alen = rng.integers(0, 90, size=k)
    alen[rng.integers(0, k, size=200)] = 0
    short = np.flatnonzero(alen <= 6)
    Ap, Ai, Ax = _ragged(rng, m, k, alen)
    blen = rng.integers(0, 40, size=n)
    blen[:6] = [300, 280, 1, 0, 513, 2]
    blen[6:11] = [70, 70, 60, 75, 130]                    # 2048 < P <= 4096 (largest table) and one global-bin column
    Bp, Bi, Bx = _ragged(rng, k, n, blen)
    for j in (0, 1, 4):                                   # long B columns name short A columns: P stays <= 4096
        Bi[Bp[j]:Bp[j + 1]] = rng.choice(short, size=blen[j])
    Cp, Ci, Cx = CO.multiply(m, k, n, Ap, Ai, Ax, Bp, Bi, Bx)
    _, _, Sx = CO.multiply(m, k, n, Ap, Ai, np.abs(Ax), Bp, Bi, np.abs(Bx))
    prods = np.array([int(np.sum(alen[Bi[Bp[j]:Bp[j + 1]]])) for j in range(n)])
    assert ((prods > 2048) & (prods <= 4096)).any() and prods.max() > 4096 and (prods[[0, 1, 4]] > 0).all()
    check_product(cs.cs_multiply(_host_cs(cs, m, k, Ap, Ai, Ax), _host_cs(cs, k, n, Bp, Bi, Bx)), Cp, Ci, Cx, Sx)
    A0, B0 = _host_cs(cs, m, k, Ap, Ai, Ax), _host_cs(cs, k, n, Bp, Bi, Bx)
    A0.x = None
    C0 = cs.cs_multiply(A0, B0)                           # pattern only
    assert C0.x is None and C0.p == Cp.tolist() and C0.i[:int(Cp[-1])] == Ci.tolist()


def test_multiply_narrow_columns_one_pass_kernel(cs):
    """Columns with at most 64 entries in B(:,j) that name A columns of at most 32 entries take k_sg_hash2: B columns of
    every length 0..64 (so every number of 8-entry slots per thread), A columns of 0..32 entries (all six table sizes up to
    2048 products), duplicate rows inside and across A columns, a few wide columns beside them that take the general
    kernel; values and pattern-only, against the C oracle (first-touch order bit for bit)."""
    rng = np.random.default_rng(20240612)
    m, k, n = 30000, 2500, 1300
    alen = rng.integers(0, 33, size=k)
    alen[:40] = 32
    Ap, Ai, Ax = _ragged(rng, m, k, alen)
    hot = rng.integers(0, m, size=50).astype(np.int32)           # rows shared by many columns: collisions on purpose
    sel = rng.random(Ai.size) < 0.3
    Ai[sel] = hot[rng.integers(0, 50, size=int(sel.sum()))]
    blen = np.concatenate([np.arange(65), rng.integers(0, 65, size=n - 65 - 5), [64, 64, 70, 100, 64]])
    Bp, Bi, Bx = _ragged(rng, k, n, blen)
    Bi[Bp[n - 5]:Bp[n - 4]] = rng.integers(0, 40, size=64)       # 64 entries x 32 = 2048 products: the largest narrow column
    prods = np.array([int(np.sum(alen[Bi[Bp[j]:Bp[j + 1]]])) for j in range(n)])
    assert prods[n - 5] == 2048 and (blen > 64).any() and (prods[blen <= 64] <= 2048).all()
    for lo, hi_ in ((0, 256), (256, 512), (512, 768), (768, 1024), (1024, 1536), (1536, 2048)):
        assert ((prods > lo) & (prods <= hi_) & (blen <= 64)).any()
    Cp, Ci, Cx = CO.multiply(m, k, n, Ap, Ai, Ax, Bp, Bi, Bx)
    _, _, Sx = CO.multiply(m, k, n, Ap, Ai, np.abs(Ax), Bp, Bi, np.abs(Bx))
    check_product(cs.cs_multiply(_host_cs(cs, m, k, Ap, Ai, Ax), _host_cs(cs, k, n, Bp, Bi, Bx)), Cp, Ci, Cx, Sx)
    A0, B0 = _host_cs(cs, m, k, Ap, Ai, Ax), _host_cs(cs, k, n, Bp, Bi, Bx)
    A0.x = None
    C0 = cs.cs_multiply(A0, B0)
    assert C0.x is None and C0.p == Cp.tolist() and C0.i[:int(Cp[-1])] == Ci.tolist()


@pytest.mark.parametrize("name", ALL)
def test_multiply_values_in_reference_order_are_bit_identical(cs, name, meta):
    """csx_set_option("spgemm.ordered", 1): after the pattern, C.x is recomputed with every entry's products added in
    the reference's order (csparse.py:1979-1988: first product assigned, later ones added as met, each rounded on its
    own).  The digests are those of the UNMODIFIED reference's cs_multiply (oracle/gen_golden.py)."""
    import _csx
    g = golden(name)
    A, AT = unpack(cs, g, "A"), unpack(cs, g, "AT")
    with _csx.option("spgemm.ordered", 1):
        C = cs.cs_multiply(A, AT)
        C2 = cs.cs_multiply(A, AT)
    mm = meta[name]["AAT"]
    nnz = C.p[C.n]
    assert sha(C.p, np.int64) == mm["sha_p"] and sha(C.i[:nnz], np.int64) == mm["sha_i"]
    assert sha(C.x[:nnz], np.float64) == mm["sha_x"]
    assert C.x == C2.x
    if name in SMALL:
        same_csc(C, g, "AAT", exact_x=True)


def test_multiply_reference_order_on_duplicates_hash_paths_and_a_long_column(cs):
    import _csx
    with _csx.option("spgemm.ordered", 1):
        g = golden("synthetic_20240601")            # duplicate rows inside columns of A and B, empty columns, cancellation
        for c, (m, n, nnz) in enumerate(g["cases"]):
            pre = "c%d_" % c
            A, AT = unpack(cs, g, pre + "A"), unpack(cs, g, pre + "AT")
            same_csc(cs.cs_multiply(A, AT), g, pre + "AAT", exact_x=True)
            same_csc(cs.cs_multiply(AT, A), g, pre + "ATA", exact_x=True)
        # LDS hash kernels (m > 8192), against the plain-C oracle bit for bit
        n = 20000
        Ap, Ai, Ax = synth.grand(n, 32, 31)
        Ax = Ax - 1.0                                # mixed signs: sums that cancel expose the order
        Tp, Ti, Tx = CO.transpose(n, n, Ap, Ai, Ax)
        Cp, Ci, Cx = CO.multiply(n, n, n, Ap, Ai, Ax, Tp, Ti, Tx)
        C = cs.cs_multiply(_host_cs(cs, n, n, Ap, Ai, Ax), _host_cs(cs, n, n, Tp, Ti, Tx))
        assert C.p == Cp.tolist() and C.i[:Cp[-1]] == Ci.tolist()
        assert np.asarray(C.x[:Cp[-1]]).tobytes() == Cx.tobytes()
        # one column with 24 000 products and ~16 000 distinct rows: the dense map in memory
        n = 30000
        Ap, Ai, Ax = synth.grand(n, 8, 77)
        Ax = Ax - 1.0
        rng = np.random.default_rng(1)
        sel = np.sort(rng.choice(n, 3000, replace=False)).astype(np.int32)
        rest = rng.integers(0, n, size=(n - 1) * 2).astype(np.int32)
        Bi = np.concatenate([sel, rest])
        Bp = np.concatenate([[0], 3000 + 2 * np.arange(n)]).astype(np.int32)
        Bx = rng.uniform(-1.5, 1.5, len(Bi))
        Cp, Ci, Cx = CO.multiply(n, n, n, Ap, Ai, Ax, Bp, Bi, Bx)
        assert Cp[1] - Cp[0] > 2048
        C = cs.cs_multiply(_host_cs(cs, n, n, Ap, Ai, Ax), _host_cs(cs, n, n, Bp, Bi, Bx))
        assert C.p == Cp.tolist() and C.i[:Cp[-1]] == Ci.tolist()
        assert np.asarray(C.x[:Cp[-1]]).tobytes() == Cx.tobytes()


def test_multiply_in_column_chunks_gives_the_same_matrix(cs):
    """spgemm.chunks >= 2 (opt-in): columns hashed in ascending chunks, each chunk compacted on a second stream behind the
    previous one's end; C.p / C.i must be what the unchunked path gives, C.x equal to rounding."""
    import _csx
    n = 40000
    Ap, Ai, Ax = synth.grand(n, 24, 5)
    Tp, Ti, Tx = CO.transpose(n, n, Ap, Ai, Ax)
    A, AT = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax)), cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    C0 = cs.cs_multiply(A, AT)
    for chunks in (2, 7):
        with _csx.option("spgemm.chunks", chunks):
            C1 = cs.cs_multiply(A, AT)
        assert C1.p == C0.p and C1.i == C0.i and C1.nzmax == C0.nzmax
        assert np.max(np.abs(np.asarray(C1.x) - np.asarray(C0.x)) / np.abs(np.asarray(C0.x))) < 1e-13


def test_multiply_in_column_chunks_while_the_copy_overlaps(cs):
    """The same on a matrix big enough for a chunk's compaction (second stream) to still be running when the next chunk's
    counts are scanned into C.p: the copy takes a column's length from the counts, not from C.p[j + 1], which the next
    chunk's scan is rewriting at that moment (round-3 advisor finding).  p, i bit for bit, x to rounding, 2 .. 8 chunks."""
    import ctypes as C
    import _csx
    lib = _csx.lib()
    n, per_col = 120000, 32
    hA, hT = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_gen_grand_uniform(n, per_col, 77, hA))
    _csx.check(lib.csx_transpose(hA, 1, hT))

    def product():
        hC = _csx.new_handle()
        _csx.check(lib.csx_multiply(hA, hT, hC))
        m_, n_, z_, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
        _csx.check(lib.csx_csc_info(hC, m_, n_, z_, hv))
        p, i, x = np.empty(n + 1, np.int32), np.empty(z_.value, np.int32), np.empty(z_.value)
        _csx.check(lib.csx_csc_download(hC, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
        _csx.free(hC)
        return p, i, x

    p0, i0, x0 = product()
    for chunks in (2, 3, 5, 8):
        with _csx.option("spgemm.chunks", chunks):
            p1, i1, x1 = product()
        assert p1.tobytes() == p0.tobytes() and i1.tobytes() == i0.tobytes(), chunks
        assert np.max(np.abs(x1 - x0) / np.abs(x0)) < 1e-13, chunks
    _csx.free(hA)
    _csx.free(hT)
