"""Assembly / reshaping functions around the hot path on the device, against the
reference's own pipeline outputs (golden fixtures made by the unmodified reference) and the oracle:
cs_compress, cs_dupl, cs_dropzeros, cs_droptol, cs_fkeep, cs_add, cs_permute, cs_symperm."""
import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, golden_meta, unpack, same_csc
from test_gpu_parity import ALL, cs  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def meta():
    return golden_meta()


def triplet(mod, g):
    T = mod.cs_spalloc(0, 0, 1, True, True)
    for i, j, x in zip(g["T_i"].tolist(), g["T_j"].tolist(), g["T_x"].tolist()):
        assert mod.cs_entry(T, int(i), int(j), float(x))
    return T


def make_sym(mod, A):
    AT = mod.cs_transpose(A, True)
    mod.cs_fkeep(AT, lambda i, j, a, o: i != j, None)
    return mod.cs_add(A, AT, 1, 1)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("name", ALL)
def test_reference_problem_pipeline(cs, name, pinned, meta):
    """csparse_test.py's get_problem: load -> compress -> dupl -> dropzeros -> droptol -> A + A' - diag.
    The result must be the unmodified reference's `C`, bit for bit (every sum here has <= 2 terms)."""
    g = golden(name)
    T = triplet(cs, g)
    if pinned:
        cs.cs_pin(T)
    A = cs.cs_compress(T)
    assert A._lazy == pinned
    assert cs.cs_dupl(A) is True
    nz1 = cs.cs_dropzeros(A)
    nz2 = cs.cs_droptol(A, 1e-14)
    Ao = O.cs_compress(triplet(O, g))
    O.cs_dupl(Ao)
    assert nz1 == O.cs_dropzeros(Ao) and nz2 == O.cs_droptol(Ao, 1e-14)
    assert A._lazy == pinned                      # a pinned pipeline never left the device
    C = make_sym(cs, A) if meta[name]["sym"] else A
    same_csc(C, g, "C")


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_symperm_golden_and_permuted(cs, name):
    g = golden(name)
    C = unpack(cs, g, "C")
    same_csc(cs.cs_symperm(C, None, False), g, "symperm")
    Co = unpack(O, g, "C")
    rng = np.random.default_rng(5)
    pinv = rng.permutation(C.n).tolist()
    got, ref = cs.cs_symperm(C, pinv, True), O.cs_symperm(Co, pinv, True)
    assert (got.m, got.n, got.nzmax, len(got.i), len(got.x)) == (ref.m, ref.n, ref.nzmax, len(ref.i), len(ref.x))
    nnz = ref.p[ref.n]
    assert got.p == ref.p and got.i[:nnz] == ref.i[:nnz] and got.x[:nnz] == ref.x[:nnz]


def _random_csc(rng, m, n, maxlen, values=True, dup=True):
    lens = rng.integers(0, maxlen + 1, size=n)
    p = np.zeros(n + 1, dtype=np.int64)
    p[1:] = np.cumsum(lens)
    nnz = int(p[-1])
    i = rng.integers(0, m, size=nnz)
    if not dup:
        for j in range(n):
            k = int(lens[j])
            if k:
                i[p[j]:p[j + 1]] = rng.choice(m, size=min(k, m), replace=False)[:k] if k <= m else i[p[j]:p[j + 1]]
    x = rng.uniform(-1, 1, size=nnz)
    x[rng.random(nnz) < 0.1] = 0.0                   # explicit zeros
    x[rng.random(nnz) < 0.05] *= 1e-9                # tiny entries
    return p.tolist(), i.tolist(), (x.tolist() if values else None)


def _mk(mod, m, n, p, i, x):
    A = mod.cs_spalloc(m, n, max(len(i), 1), x is not None, False)
    A.p, A.i = list(p), list(i) if i else [0]
    A.x = None if x is None else (list(x) if x else [0.0])
    return A


def _same(C, R, exact=True):
    assert (C.m, C.n, C.nz, C.nzmax, len(C.i)) == (R.m, R.n, R.nz, R.nzmax, len(R.i))
    nnz = R.p[R.n]
    assert list(C.p) == list(R.p) and list(C.i[:nnz]) == list(R.i[:nnz])
    assert (C.x is None) == (R.x is None)
    if R.x is not None:
        assert len(C.x) == len(R.x)
        if exact:
            assert list(C.x[:nnz]) == list(R.x[:nnz])
        elif nnz:
            a, b = np.asarray(C.x[:nnz]), np.asarray(R.x[:nnz])
            assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))


@pytest.mark.parametrize("m,n,maxlen", [(9000, 40, 300), (60, 70, 40), (20000, 500, 60), (5, 3, 2), (4, 0, 0)])
def test_add_dupl_drop_permute_against_oracle(cs, m, n, maxlen):
    rng = np.random.default_rng(m * 7 + n)
    a = _random_csc(rng, m, n, maxlen)
    b = _random_csc(rng, m, n, max(maxlen // 2, 0))
    A, B, Ao, Bo = _mk(cs, m, n, *a), _mk(cs, m, n, *b), _mk(O, m, n, *a), _mk(O, m, n, *b)
    # duplicates inside a column make some sums longer than two terms: rounding-level agreement
    _same(cs.cs_add(A, B, 0.75, -2.5), O.cs_add(Ao, Bo, 0.75, -2.5), exact=False)
    _same(cs.cs_add(A, A, 1.0, 1.0), O.cs_add(Ao, Ao, 1.0, 1.0), exact=False)
    # pattern-only operands give a pattern-only sum
    Ap_, Bp_ = _mk(cs, m, n, a[0], a[1], None), _mk(O, m, n, a[0], a[1], None)
    _same(cs.cs_add(Ap_, B, 1, 1), O.cs_add(Bp_, Bo, 1, 1))
    assert cs.cs_add(A, _mk(cs, m + 1, n, *b), 1, 1) is None
    # permute: rows renamed, columns gathered, storage order kept
    pinv, q = rng.permutation(m).tolist(), rng.permutation(n).tolist()
    _same(cs.cs_permute(A, pinv, q, True), O.cs_permute(Ao, pinv, q, True))
    _same(cs.cs_permute(A, None, q, False), O.cs_permute(Ao, None, q, False))
    _same(cs.cs_permute(A, pinv, None, True), O.cs_permute(Ao, pinv, None, True))
    # in-place family
    assert cs.cs_dupl(A) is True and O.cs_dupl(Ao) is True
    _same(A, Ao, exact=False)
    assert cs.cs_dropzeros(A) == O.cs_dropzeros(Ao)
    _same(A, Ao, exact=False)
    assert cs.cs_droptol(A, 1e-6) == O.cs_droptol(Ao, 1e-6)
    _same(A, Ao, exact=False)
    keep = lambda i, j, aij, other: (i + j) % other != 0          # noqa: E731
    assert cs.cs_fkeep(A, keep, 3) == O.cs_fkeep(Ao, keep, 3)
    _same(A, Ao, exact=False)


def test_add_without_duplicates_is_bit_exact(cs):
    rng = np.random.default_rng(11)
    m, n = 12000, 300
    a, b = _random_csc(rng, m, n, 50, dup=False), _random_csc(rng, m, n, 50, dup=False)
    _same(cs.cs_add(_mk(cs, m, n, *a), _mk(cs, m, n, *b), 3.0, -0.125),
          O.cs_add(_mk(O, m, n, *a), _mk(O, m, n, *b), 3.0, -0.125), exact=True)


def test_bad_arguments(cs):
    g = golden("t1")
    A = unpack(cs, g, "A")
    T = cs.cs_spalloc(0, 0, 1, True, True)
    assert cs.cs_add(A, None, 1, 1) is None and cs.cs_add(T, A, 1, 1) is None
    assert cs.cs_dupl(T) is False and cs.cs_dropzeros(T) == -1 and cs.cs_droptol(None, 1.0) == -1
    assert cs.cs_permute(T, None, None, True) is None and cs.cs_symperm(None, None, True) is None
    assert cs.cs_compress(A) is None and cs.cs_fkeep(A, None, None) == -1
    with pytest.raises(IndexError):
        cs.cs_permute(A, [0, 1], None, True)          # pinv shorter than m


@pytest.mark.parametrize("name", ALL)
def test_norm_on_device_has_the_reference_bits(cs, name, meta):
    g = golden(name)
    A = unpack(cs, g, "A")
    host = cs.cs_norm(A)
    cs.cs_pin(A)
    assert cs.cs_norm(A) == host == meta[name]["normA"]          # meta: the unmodified reference's cs_norm
    P = unpack(cs, g, "A")
    P.x = None
    assert cs.cs_norm(cs.cs_pin(P)) == -1 and cs.cs_norm(None) == -1


@pytest.mark.parametrize("name", ["west0067", "ash219", "lp_afiro", "bcsstk16"])
def test_col_block_is_the_oracle_column_slice(cs, name):
    """csx_csc_col_block (the unit of a column-sharded SpMV): columns [first, first + count) of A as an
    m x count matrix -- p rebased to 0, i and x the slice, bit for bit -- for every split the strong-scaling
    partition makes at 1 / 2 / 3 / 8 ranks, and empty and full ranges; the blocks' SpMVs sum to the unsharded one."""
    import _csx
    import shard
    lib = _csx.lib()
    g = golden(name)
    A = cs.cs_pin(unpack(cs, g, "A"))
    Ap, nnz = np.asarray(g["A_p"]), int(g["A_p"][-1])
    Ai, Ax = np.asarray(g["A_i"]), np.asarray(g["A_x"])
    m, n = A.m, A.n
    x = 1.0 + np.arange(n) / n
    yfull = np.zeros(m)
    dy = cs.dvec(m)
    assert cs.cs_gaxpy(A, cs.dvec(x), dy, cs.GAXPY_EXACT)
    yfull = dy.numpy()
    for world in (1, 2, 3, 8):
        ysum = np.zeros(m)
        for r in range(world):
            first, count = shard.strong_block(r, world, n)
            h = _csx.new_handle()
            _csx.check(lib.csx_csc_col_block(A._dev.handle, first, count, h))
            mm, nn, zz, hv = (_csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int())
            _csx.check(lib.csx_csc_info(h, mm, nn, zz, hv))
            lo, hi = int(Ap[first]), int(Ap[first + count])
            assert (mm.value, nn.value, zz.value, hv.value) == (m, count, hi - lo, 1)
            p, i, v = np.empty(count + 1, np.int32), np.empty(max(hi - lo, 1), np.int32), np.empty(max(hi - lo, 1))
            _csx.check(lib.csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(v)))
            assert (p == Ap[first:first + count + 1] - lo).all()
            assert (i[:hi - lo] == Ai[lo:hi]).all() and v[:hi - lo].tobytes() == Ax[lo:hi].tobytes()
            part = cs.dvec(m)
            xs = cs.dvec(x[first:first + count] if count else np.zeros(1))     # this rank's slice of x
            _csx.check(lib.csx_gaxpy(h, xs.handle, part.handle, cs.GAXPY_EXACT))
            ysum += part.numpy()
            _csx.free(h)
        scale = np.abs(yfull) + 1e-300
        assert np.max(np.abs(ysum - yfull) / scale) < 1e-12, world
    h = _csx.new_handle()
    assert lib.csx_csc_col_block(A._dev.handle, n - 1, 2, h) == _csx.EINVAL
    assert lib.csx_csc_col_block(A._dev.handle, -1, 1, h) == _csx.EINVAL
    _csx.check(lib.csx_csc_col_block(A._dev.handle, n, 0, h))          # empty block at the end is legal
    _csx.free(h)
