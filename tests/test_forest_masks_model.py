"""The ALGORITHM of the sparse-forest path (csparse.py_amd/csrc/csx_cholclique.hip: k_forest_mark, k_forest_symbolic and the
compacted store of k_chol_clique), restated in a few lines of Python on integers and checked against the plain-C port of
cs_schol / cs_chol on random block patterns.  Runs without a GPU: it pins the rule the kernels implement -- blocks from a
reverse running minimum of the smallest upper row, symbolic elimination on 64-bit row masks, a column's place in L by
popcounts -- not the kernels (tests/test_gpu_cholclique.py does that on the device)."""
import numpy as np

import c_oracle as CO


def _blocks(u):
    """k starts a block iff no column j >= k reaches above k: min_{j >= k} u[j] == k."""
    smin = np.minimum.accumulate(u[::-1])[::-1]
    return np.nonzero(smin == np.arange(len(u)))[0]


def _symbolic(n, Ap, Ai):
    """parent, counts and the column masks of L, block by block, on row masks (bit r = row start + r)."""
    Ap, Ai = [int(v) for v in Ap], [int(v) for v in Ai]          # Python integers: the masks have 64 bits
    u = np.array([min([i for i in Ai[Ap[k]:Ap[k + 1]] if i <= k] + [k]) for k in range(n)])
    starts = [int(v) for v in _blocks(u)] + [n]
    parent, count, colmask = np.full(n, -1), np.zeros(n, int), [0] * n
    for c0, c1 in zip(starts[:-1], starts[1:]):
        bs = c1 - c0
        assert bs <= 64
        mask = []
        for r in range(bs):                      # lane r: the upper part of column c0 + r of A is row r of L's pattern
            m = 1 << r
            for i in Ai[Ap[c0 + r]:Ap[c0 + r + 1]]:
                if c0 <= i <= c0 + r:
                    m |= 1 << (i - c0)
            mask.append(m)
        for j in range(bs):
            cj = sum(1 << r for r in range(bs) if mask[r] >> j & 1)          # the ballot of bit j
            colmask[c0 + j] = cj
            for r in range(j + 1, bs):
                if mask[r] >> j & 1:
                    mask[r] |= cj & ((1 << r) - 1) & ~((2 << j) - 1)          # the column's rows between j and r
            under = cj & ~((2 << j) - 1)
            count[c0 + j] = 1 + bin(under).count("1")
            if under:
                parent[c0 + j] = c0 + (under & -under).bit_length() - 1
    return starts, parent, count, colmask


def _random_forest(rng, sizes, density):
    cols, a = [], 0
    for bs in sizes:
        K = np.triu(rng.uniform(size=(bs, bs)) < density) | np.eye(bs, dtype=bool)
        K = K | K.T
        for c in range(bs):
            cols.append(np.nonzero(K[:, c])[0] + a)
        a += bs
    Ap = np.zeros(a + 1, np.int32)
    Ap[1:] = np.cumsum([len(c) for c in cols])
    return int(a), Ap, np.concatenate(cols).astype(np.int32)


def test_row_mask_elimination_gives_the_reference_tree_and_counts():
    rng = np.random.default_rng(17)
    for density in (0.05, 0.15, 0.4, 1.0):
        sizes = list(rng.integers(1, 65, 40)) + [64, 1]
        n, Ap, Ai = _random_forest(rng, sizes, density)
        parent, cp = CO.schol(n, Ap, Ai)
        starts, par, count, colmask = _symbolic(n, Ap, Ai)
        assert par.tolist() == parent.tolist()
        assert np.concatenate([[0], np.cumsum(count)]).tolist() == cp.tolist()
        # every elimination tree lies inside one block, and the blocks are no coarser than the generator's
        assert set(int(v) for v in np.concatenate([[0], np.cumsum(sizes)])) <= set(starts)
        for k in range(n):
            if parent[k] >= 0:
                b = np.searchsorted(starts, k, side="right") - 1
                assert starts[b] <= parent[k] < starts[b + 1]


def test_a_column_of_L_is_its_mask_and_an_entrys_place_a_popcount():
    rng = np.random.default_rng(3)
    sizes = list(rng.integers(2, 50, 25))
    n, Ap, Ai = _random_forest(rng, sizes, 0.2)
    Ax = np.zeros(len(Ai))
    for k in range(n):                                           # diagonally dominant values on that pattern
        for p in range(Ap[k], Ap[k + 1]):
            Ax[p] = 64.0 if Ai[p] == k else -1.0 / (1 + abs(int(Ai[p]) - k))
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    starts, par, count, colmask = _symbolic(n, Ap, Ai)
    for c0, c1 in zip(starts[:-1], starts[1:]):
        for g in range(c0, c1):
            cg = colmask[g]
            rows = [c0 + r for r in range(c1 - c0) if cg >> r & 1]
            assert rows == Li[Lp[g]:Lp[g + 1]].tolist()          # diagonal first, rows ascending
            for r in rows:                                        # the store's address: popcount of the mask below the lane
                assert Li[Lp[g] + bin(cg & ((1 << (r - c0)) - 1)).count("1")] == r


def test_the_clique_rule_is_an_iff():
    """k_clique_mark's rule -- u[0] = 0 and u[k] in {k, u[k - 1]} -- holds exactly when the elimination forest is a set of
    cliques on consecutive columns: every tree a chain of consecutive columns and every column of L full inside its block
    (parent[k] = k + 1 or -1, count[k] = block end - k), checked against the plain-C port's cs_schol on random patterns, some
    built to satisfy the rule (first row / column of every block full), some arbitrary."""
    rng = np.random.default_rng(29)
    seen = {True: 0, False: 0}
    for trial in range(60):
        sizes = [int(v) for v in rng.integers(1, 20, 12)]
        cols, a = [], 0
        full_first = trial % 2 == 0
        for bs in sizes:
            K = np.triu(rng.uniform(size=(bs, bs)) < rng.choice([0.2, 0.6, 1.0])) | np.eye(bs, dtype=bool)
            if full_first:
                K[0, :] = True
            K = K | K.T
            for c in range(bs):
                cols.append(np.nonzero(K[:, c])[0] + a)
            a += bs
        n = int(a)
        Ap = np.zeros(n + 1, np.int32)
        Ap[1:] = np.cumsum([len(c) for c in cols])
        Ai = np.concatenate(cols).astype(np.int32)
        u = [min([int(i) for i in Ai[Ap[k]:Ap[k + 1]] if i <= k] + [k]) for k in range(n)]
        rule = u[0] == 0 and all(u[k] == k or u[k] == u[k - 1] for k in range(1, n))
        parent, cp = CO.schol(n, Ap, Ai)
        count = np.diff(cp)
        # cliques on consecutive columns, read off the reference's tree and counts
        ends = [k for k in range(n) if parent[k] < 0]
        starts = [0] + [e + 1 for e in ends[:-1]]
        cliques = all(parent[k] in (-1, k + 1) for k in range(n)) and all(
            count[k] == e - k + 1 for s0, e in zip(starts, ends) for k in range(s0, e + 1))
        assert rule == cliques, (trial, sizes)
        seen[rule] += 1
    assert seen[True] >= 20 and seen[False] >= 10
