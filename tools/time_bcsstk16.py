#!/usr/bin/env python3
"""bcsstk16 (the reference's own test matrix, csparse_test.py:525): cs_chol and the four triangular solves on its
factor, HIP-event times, with the plain-C oracle on one host core beside them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx, csparse as cs, c_oracle as CO
from conftest import golden, unpack
_csx.init(0); lib = _csx.lib()
g = golden("bcsstk16")
C = cs.cs_pin(unpack(cs, g, "C"))
n = C.n
for order in (0, 1):
    S = cs.cs_schol(order, C)
    N = cs.cs_chol(C, S); _csx.sync()
    t0 = time.perf_counter(); N = cs.cs_chol(C, S); _csx.sync(); t_chol = time.perf_counter() - t0
    L = cs.cs_pin(N.L)
    lnz = S.lnz
    out = {"order": order, "lnz": lnz, "chol_ms": round(t_chol * 1e3, 2)}
    for k in (1, 64):
        B = np.ones((n, k)) if k > 1 else np.ones(n)
        for name, fn in (("lsolve", cs.cs_lsolve), ("ltsolve", cs.cs_ltsolve)):
            X = cs.dvec(B); fn(L, X); _csx.sync()
            plan = L._dev.plans[cs.TRI_L if name == "lsolve" else cs.TRI_LT]
            with _csx.Timer() as tm:
                for _ in range(5):
                    lib.csx_tri_solve(plan, X.handle, k)
            out["%s_k%d_ms" % (name, k)] = round(tm.ms / 5, 3)
    print(out)
p, i, x = g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"]
parent, cp = CO.schol(n, p, i)
t0 = time.perf_counter(); Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp); t1 = time.perf_counter()
b = np.ones(n)
t2 = time.perf_counter(); y = CO.lsolve(n, Lp, Li, Lx, b); t3 = time.perf_counter(); z = CO.ltsolve(n, Lp, Li, Lx, y); t4 = time.perf_counter()
print({"plain_c_one_core": {"chol_ms": round((t1 - t0) * 1e3, 2), "lsolve_ms": round((t3 - t2) * 1e3, 3), "ltsolve_ms": round((t4 - t3) * 1e3, 3)}})
