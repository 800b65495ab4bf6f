import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, scipy.sparse as sp
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
g = int(sys.argv[1]); k = int(sys.argv[2]); order = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
B = cs.dvec(np.ones((n, k)))
for _ in range(4):
    F.solve(B)
_csx.sync()
