"""Where cs_schol(1, A) spends its time on a grid Laplacian (host side, cProfile)."""
import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import scipy.sparse as sp
import _csx, csparse as cs
_csx.init(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 700
n = g * g
T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc()
A.sort_indices()
M = cs.cs_spalloc(n, n, A.nnz, True, False)
M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
cs.cs_pin(M)
S = cs.cs_schol(1, M)
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter(); S = cs.cs_schol(1, M); _csx.sync(); dt = time.perf_counter() - t0
pr.disable()
print("cs_schol(1) %.3f s" % dt)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
