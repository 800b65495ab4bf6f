"""One large connected problem through the supernodal path: a g x g grid Laplacian at order 1 (default g = 1500: n = 2.25 M),
rounding-equal solve of 8 right-hand sides, residual and agreement with the exact order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, scipy.sparse as sp
import _csx, csparse as cs
_csx.init(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n = g * g
T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc(); A.sort_indices()
M = cs.cs_spalloc(n, n, A.nnz, True, False)
M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
cs.cs_pin(M)
rng = np.random.default_rng(5)
B = rng.uniform(-1, 1, size=(n, 8))
out = {"n": n}
for exact in (False, True):
    t0 = time.perf_counter(); F = cs.cholsol_factor(M, 1, exact=exact); _csx.sync(); out["factor_s_%s" % exact] = round(time.perf_counter() - t0, 2)
    X = cs.dvec(B)
    F.solve(X); _csx.sync()
    X2 = cs.dvec(B)
    with _csx.Timer() as tm:
        F.solve(X2)
    out["solve_ms_%s" % ("exact" if exact else "rounding_equal")] = round(tm.ms, 2)
    Xn = X2.numpy().reshape(n, 8)
    out["residual_%s" % ("exact" if exact else "rounding_equal")] = float(np.max(np.abs(A @ Xn - B)))
    if exact:
        out["max_diff_between_orders"] = float(np.max(np.abs(Xn - keep)))
    keep = Xn
    out["lnz"] = int(F.symbolic.lnz)
    print(out, flush=True)
