"""Exact (default) order on order-1 factors: how the time splits between L x = b and L' x = b."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, scipy.sparse as sp
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
for g in (0, 300, 700):
    if g == 0:
        M = cs.cs_pin(unpack(cs, golden("bcsstk16"), "C")); n = M.n
    else:
        n = g * g
        T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
        A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc(); A.sort_indices()
        M = cs.cs_spalloc(n, n, A.nnz, True, False)
        M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
        cs.cs_pin(M)
    F = cs.cholsol_factor(M, 1, exact=True)
    L = F.L
    for k in (1, 64):
        B = cs.dvec(np.ones((n, k)) if k > 1 else np.ones(n))
        out = {}
        for name, fn in (("lsolve", cs.cs_lsolve), ("ltsolve", cs.cs_ltsolve)):
            fn(L, B); _csx.sync()
            with _csx.Timer() as tm:
                for _ in range(3):
                    fn(L, B)
            out[name] = round(tm.ms / 3, 3)
        print("grid %d k %d: %s" % (g, k, out), flush=True)
