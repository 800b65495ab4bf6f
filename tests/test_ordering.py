"""csx_order_nd_host (the order = 1 ordering): a valid permutation, deterministic, and doing its job -- on the
reference's structural matrices the elimination tree of P A P' must be much shallower than the natural one
without blowing up the fill."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as CO
import csparse_oracle as O
from conftest import golden, unpack


def nd(n, p, i):
    import _csx
    lib = _csx.load()
    perm = np.empty(max(n, 1), dtype=np.int32)
    assert lib.csx_order_nd_host(n, _csx.pi(p), _csx.pi(i), _csx.pi(perm)) == 0
    return perm[:n]


def tree_height(parent):
    depth = np.zeros(len(parent), dtype=np.int64)
    for v in range(len(parent) - 1, -1, -1):        # parents have larger indices
        if parent[v] >= 0:
            depth[v] = depth[parent[v]] + 1
    return int(depth.max()) + 1 if len(parent) else 0


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_nested_dissection_on_reference_matrices(name):
    g = golden(name)
    Cm = unpack(O, g, "C")
    n = Cm.n
    p, i = np.asarray(Cm.p, np.int32), np.asarray(Cm.i[:Cm.p[n]], np.int32)
    perm = nd(n, p, i)
    assert sorted(perm.tolist()) == list(range(n))
    assert (nd(n, p, i) == perm).all()                                     # deterministic
    pinv = O.cs_pinv(perm.tolist(), n)
    C2 = O.cs_symperm(Cm, pinv, False)
    p2, i2 = np.asarray(C2.p, np.int32), np.asarray(C2.i[:C2.p[n]], np.int32)
    par_nat, cp_nat = CO.schol(n, p, i)
    par_nd, cp_nd = CO.schol(n, p2, i2)
    h_nat, h_nd = tree_height(par_nat), tree_height(par_nd)
    if name == "bcsstk16":                                                # (bcsstk01 is one 48-vertex leaf)
        assert h_nd * 3 < h_nat                                            # 4810 levels naturally
        assert cp_nd[n] < 2.5 * cp_nat[n]                                  # fill stays in the same league
    print(name, "levels", h_nat, "->", h_nd, "lnz", int(cp_nat[n]), "->", int(cp_nd[n]))


def test_ordering_edge_cases():
    import _csx
    lib = _csx.load()
    # empty, diagonal, two components, a path, a dense block
    for n, edges in ((0, []), (5, []), (6, [(0, 1), (1, 2), (3, 4)]), (300, [(k, k + 1) for k in range(299)]),
                     (40, [(a, b) for a in range(40) for b in range(a)])):
        cols = [[] for _ in range(n)]
        for a, b in edges:
            cols[a].append(b)
            cols[b].append(a)
        for j in range(n):
            cols[j].append(j)
        p = np.zeros(n + 1, np.int32)
        p[1:] = np.cumsum([len(c) for c in cols]) if n else []
        i = np.asarray([r for c in cols for r in c], dtype=np.int32)
        perm = nd(n, p, i)
        assert sorted(perm.tolist()) == list(range(n))
    bad_p, bad_i, out = np.asarray([0, 1], np.int32), np.asarray([7], np.int32), np.zeros(1, np.int32)
    assert lib.csx_order_nd_host(1, _csx.pi(bad_p), _csx.pi(bad_i), _csx.pi(out)) == _csx.EINVAL
