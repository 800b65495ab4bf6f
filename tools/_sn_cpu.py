"""Is a supernodal solve bound by the host's launch rate?  Enqueue time (F.solve returns) against completion time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, scipy.sparse as sp
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
for g, order in ((0, 1), (300, 1), (700, 1), (300, 0)):
    if g == 0:
        M = cs.cs_pin(unpack(cs, golden("bcsstk16"), "C")); n = M.n
    else:
        n = g * g
        T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
        A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc(); A.sort_indices()
        M = cs.cs_spalloc(n, n, A.nnz, True, False)
        M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
        cs.cs_pin(M)
    F = cs.cholsol_factor(M, order, exact=False)
    B = cs.dvec(np.ones((n, 8)))
    F.solve(B); _csx.sync()
    for graph in (0, 1):
        _csx.check(_csx.lib().csx_set_option(b"tri.graph", graph))
        F.solve(B); _csx.sync()
        enq, tot = [], []
        for _ in range(5):
            t0 = time.perf_counter(); F.solve(B); t1 = time.perf_counter(); _csx.sync(); t2 = time.perf_counter()
            enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
        print("grid %d order %d graph %d: enqueue %.3f ms, complete %.3f ms" % (g, order, graph, min(enq), min(tot)), flush=True)
    _csx.check(_csx.lib().csx_set_option(b"tri.graph", 0))
