"""One-shot drop-in calls on bcsstk16 (the reference's own test matrix): cs_cholsol(0, C, b) from a pinned matrix and a
device vector, every phase timed, against the plain-C port on one host core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx, csparse as cs, c_oracle as CO
from conftest import golden, unpack
_csx.init(0)
g = golden("bcsstk16")
C = cs.cs_pin(unpack(cs, g, "C"))
n = C.n
b = g["b"]
def T(f):
    _csx.sync(); t0 = time.perf_counter(); r = f(); _csx.sync(); return r, (time.perf_counter() - t0) * 1e3
for rep in range(3):
    S, t_s = T(lambda: cs.cs_schol(0, C))
    N, t_c = T(lambda: cs.cs_chol(C, S))
    x = cs.dvec(b.copy())
    _, t_l = T(lambda: cs.cs_lsolve(N.L, x))
    _, t_lt = T(lambda: cs.cs_ltsolve(N.L, x))
    x2 = cs.dvec(b.copy())
    _, t_l2 = T(lambda: cs.cs_lsolve(N.L, x2))
    _, t_lt2 = T(lambda: cs.cs_ltsolve(N.L, x2))
    xb = cs.dvec(b.copy())
    _, t_all = T(lambda: cs.cs_cholsol(0, C, xb))
    print({"rep": rep, "schol_ms": round(t_s, 2), "chol_ms": round(t_c, 2), "lsolve_first_ms": round(t_l, 2),
           "ltsolve_first_ms": round(t_lt, 2), "lsolve_again_ms": round(t_l2, 2), "ltsolve_again_ms": round(t_lt2, 2),
           "cs_cholsol_ms": round(t_all, 2)}, flush=True)
p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
t0 = time.perf_counter()
parent, cp = CO.schol(n, p, i); Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
z = CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b))
print({"plain_c_one_core_cholsol_ms": round((time.perf_counter() - t0) * 1e3, 2)})
