"""cs_cholsol pieces on a 2-D grid Laplacian (5-point stencil + 4 I), natural order and the order-1 ordering
(nested dissection): a connected problem with a bushy elimination tree, unlike the block forest of the benchmark."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import scipy.sparse as sp
import _csx, csparse as cs
_csx.init(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 300
orders = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
only_rhs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = g * g
T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc()
A.sort_indices()
M = cs.cs_spalloc(n, n, A.nnz, True, False)
M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
cs.cs_pin(M)
for order in orders:
    t0 = time.perf_counter(); S = cs.cs_schol(order, M); _csx.sync(); t1 = time.perf_counter()
    if S is None:
        print({"order": order, "schol": None}); continue
    N = cs.cs_chol(M, S); _csx.sync(); t2 = time.perf_counter()
    if N is None:
        print({"order": order, "lnz": S.lnz, "chol": None}); continue
    b = np.ones(n)
    x = b.tolist()
    t3 = time.perf_counter(); ok = cs.cs_cholsol(order, M, x); _csx.sync(); t4 = time.perf_counter()
    r = A @ np.asarray(x) - b
    print({"grid": g, "n": n, "order": order, "lnz": int(S.lnz), "schol_s": round(t1 - t0, 3), "chol_s": round(t2 - t1, 3),
           "cholsol_all_s": round(t4 - t3, 3), "residual_inf": float(np.max(np.abs(r)))}, flush=True)
    for exact in (True, False):
        F = cs.cholsol_factor(M, order, exact=exact)
        for k in (1, 64):
            if only_rhs and k != only_rhs:
                continue
            B = cs.dvec(np.ones((n, k)) if k > 1 else np.ones(n))
            F.solve(B); _csx.sync()
            t0 = time.perf_counter(); F.solve(B); _csx.sync(); dt = time.perf_counter() - t0
            print({"order": order, "exact": exact, "nrhs": k, "solve_s": round(dt, 4)}, flush=True)
