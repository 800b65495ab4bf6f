"""Host (numpy) twins of the device generators in csparse.py_amd/csrc/csx_gen.hip:
same counter-based hash, same operation order, bit-identical output."""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def mix64(z):
    with np.errstate(over="ignore"):
        z = (np.asarray(z, dtype=np.uint64) + _M1)
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def hash2(seed, c):
    return mix64(np.uint64(seed) ^ mix64(c))


def unit(h):
    return (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def grand(n, per_col, seed):
    """G-rand: (Ap, Ai, Ax) int32/int32/float64."""
    nnz = n * per_col
    e = np.arange(nnz, dtype=np.uint64)
    k = (e % np.uint64(per_col)).astype(np.int64)
    W = n // per_col
    Wk = np.where(k == per_col - 1, n - (per_col - 1) * W, W).astype(np.uint64)
    Ai = (k * W + (hash2(seed, e) % Wk).astype(np.int64)).astype(np.int32)
    Ax = 0.5 + unit(hash2(seed + 1, e))
    Ap = (np.arange(n + 1, dtype=np.int64) * per_col).astype(np.int32)
    return Ap, Ai, Ax


def grand_uniform(n, per_col, seed):
    """G-rand with per_col DISTINCT uniform rows per column, ascending (csx_gen_grand_uniform's twin)."""
    assert per_col <= 64 and 2 * per_col <= n
    nnz = n * per_col
    e = np.arange(nnz, dtype=np.uint64)
    t = (e % np.uint64(per_col)).reshape(n, per_col)
    with np.errstate(over="ignore"):
        r = (hash2(seed, e * np.uint64(64)) % np.uint64(n)).reshape(n, per_col)
    key = (r << np.uint64(12)) | (t << np.uint64(6))
    key.sort(axis=1)
    rows = key >> np.uint64(12)
    bad = np.nonzero((rows[:, 1:] == rows[:, :-1]).any(axis=1))[0]
    for j in bad:                      # a few columns per thousand: redraw the later of two equal rows
        k = key[j].copy()
        for _ in range(64):
            k.sort()
            rr = k >> np.uint64(12)
            dup = np.zeros(per_col, dtype=bool)
            dup[1:] = rr[1:] == rr[:-1]
            if not dup.any():
                break
            tt = (k[dup] >> np.uint64(6)) & np.uint64(63)
            aa = (k[dup] & np.uint64(63)) + np.uint64(1)
            with np.errstate(over="ignore"):
                et = np.uint64(j) * np.uint64(per_col) + tt
                k[dup] = ((hash2(seed, et * np.uint64(64) + aa) % np.uint64(n)) << np.uint64(12)) | (tt << np.uint64(6)) | aa
        key[j] = k
    Ai = (key >> np.uint64(12)).astype(np.int32).reshape(-1)
    Ax = 0.5 + unit(hash2(seed + 1, e))
    Ap = (np.arange(n + 1, dtype=np.int64) * per_col).astype(np.int32)
    return Ap, Ai, Ax


def grand_uniform_columns(n, per_col, seed, j0, count):
    """Columns [j0, j0 + count) of grand_uniform(n, per_col, seed) without building the rest: (Ai, Ax) of the window.
    (Every column is a pure function of its own entry numbers, so a window of the full-size matrix can be checked.)"""
    e = np.arange(j0 * per_col, (j0 + count) * per_col, dtype=np.uint64)
    t = (e % np.uint64(per_col)).reshape(count, per_col)
    with np.errstate(over="ignore"):
        r = (hash2(seed, e * np.uint64(64)) % np.uint64(n)).reshape(count, per_col)
    key = (r << np.uint64(12)) | (t << np.uint64(6))
    key.sort(axis=1)
    rows = key >> np.uint64(12)
    bad = np.nonzero((rows[:, 1:] == rows[:, :-1]).any(axis=1))[0]
    for jj in bad:
        k = key[jj].copy()
        for _ in range(64):
            k.sort()
            rr = k >> np.uint64(12)
            dup = np.zeros(per_col, dtype=bool)
            dup[1:] = rr[1:] == rr[:-1]
            if not dup.any():
                break
            tt = (k[dup] >> np.uint64(6)) & np.uint64(63)
            aa = (k[dup] & np.uint64(63)) + np.uint64(1)
            with np.errstate(over="ignore"):
                et = np.uint64(j0 + jj) * np.uint64(per_col) + tt
                k[dup] = ((hash2(seed, et * np.uint64(64) + aa) % np.uint64(n)) << np.uint64(12)) | (tt << np.uint64(6)) | aa
        key[jj] = k
    Ai = (key >> np.uint64(12)).astype(np.int32).reshape(-1)
    Ax = 0.5 + unit(hash2(seed + 1, e))
    return Ai, Ax


def gspd(nblocks, bs, seed):
    """G-spd: block-diagonal SPD, dense bs-by-bs blocks."""
    n = nblocks * bs
    c = np.arange(nblocks * bs * bs, dtype=np.uint64)
    R = (-1.0 + 2.0 * unit(hash2(seed, c))).reshape(nblocks, bs, bs)   # R[b][r][k]
    acc = np.zeros((nblocks, bs, bs))
    for k in range(bs):  # ascending k, separately rounded multiply and add
        acc = acc + R[:, :, k][:, :, None] * R[:, :, k][:, None, :]
    B = acc / float(bs)
    idx = np.arange(bs)
    B[:, idx, idx] = B[:, idx, idx] + float(bs)
    # column-major inside each block: entry q = ((b*bs + c)*bs + r) holds B[b][r][c]
    Ax = np.ascontiguousarray(np.transpose(B, (0, 2, 1))).reshape(-1)
    Ai = (np.arange(nblocks)[:, None, None] * bs + np.arange(bs)[None, None, :] +
          np.zeros((1, bs, 1), dtype=np.int64)).astype(np.int32).reshape(-1)
    Ap = (np.arange(n + 1, dtype=np.int64) * bs).astype(np.int32)
    return Ap, Ai, Ax


def vec(length, seed, lo, hi):
    i = np.arange(length, dtype=np.uint64)
    return lo + (hi - lo) * unit(hash2(seed, i))


def rhs(n, nrhs, col0=0):
    i = np.arange(n, dtype=np.float64)[:, None]
    r = np.arange(nrhs, dtype=np.float64)[None, :]
    return 1.0 + (i + col0 + r) / float(n)


def ragged_cliques(n_want, lo, hi, seed):
    """Block-diagonal SPD matrix whose dense blocks have sizes drawn uniformly from [lo, hi], about n_want rows (full symmetric
    storage, columns ascending): a forest of cliques of UNEQUAL sizes (north_star's "batches of independent matrices").
    Block b = R R' / s + s I.  Returns n, Ap, Ai, Ax, sizes."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi + 1, max(1, int(n_want / ((lo + hi) / 2.0))))
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(start[-1])
    Ap = np.concatenate([[0], np.cumsum(np.repeat(sizes, sizes))]).astype(np.int64)
    nnz = int(Ap[-1])
    assert nnz < 2 ** 31
    Ai = np.empty(nnz, np.int32)
    Ax = np.empty(nnz, np.float64)
    for s in range(lo, hi + 1):
        blk = np.nonzero(sizes == s)[0]
        if len(blk) == 0:
            continue
        R = rng.uniform(-1.0, 1.0, (len(blk), s, s))
        B = R @ np.transpose(R, (0, 2, 1)) / s
        B[:, np.arange(s), np.arange(s)] += s
        c0 = start[blk]
        pos = (Ap[c0][:, None, None] + (np.arange(s) * s)[None, :, None] + np.arange(s)[None, None, :]).reshape(-1)   # [block][col][row]
        Ai[pos] = (c0[:, None, None] + np.zeros((1, s, 1), np.int64) + np.arange(s)[None, None, :]).reshape(-1).astype(np.int32)
        Ax[pos] = np.transpose(B, (0, 2, 1)).reshape(-1)
    return n, Ap.astype(np.int32), Ai, Ax, sizes
