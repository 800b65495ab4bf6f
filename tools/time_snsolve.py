#!/usr/bin/env python3
"""Order-1 (nested dissection) solves on connected problems, exact order against the rounding-equal order (supernodal
schedule, csx_snsolve.hip): grid Laplacians and bcsstk16; HIP-event times of F.solve for 1 and 64 right-hand sides.
usage: time_snsolve.py [grid sizes, comma separated; 0 = bcsstk16] [order: 1 (default) or 0 = natural]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import scipy.sparse as sp
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,300,700").split(",")]
order = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for g in sizes:
    if g == 0:
        M = cs.cs_pin(unpack(cs, golden("bcsstk16"), "C")); n = M.n; name = "bcsstk16"
    else:
        n = g * g
        T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
        A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc(); A.sort_indices()
        M = cs.cs_spalloc(n, n, A.nnz, True, False)
        M.p, M.i, M.x = A.indptr.tolist(), A.indices.tolist(), A.data.tolist()
        cs.cs_pin(M); name = "grid %d x %d" % (g, g)
    for exact in (True, False):
        t0 = time.perf_counter(); F = cs.cholsol_factor(M, order, exact=exact); _csx.sync(); tf = time.perf_counter() - t0
        out = {"problem": name, "order": order, "n": n, "lnz": int(F.symbolic.lnz), "exact": exact, "factor_and_plan_s": round(tf, 3)}
        for k in (1, 8, 64):
            B = cs.dvec(np.ones((n, k)) if k > 1 else np.ones(n))
            F.solve(B); _csx.sync()
            reps = 5
            with _csx.Timer() as tm:
                for _ in range(reps):
                    F.solve(B)
            out["solve_ms_k%d" % k] = round(tm.ms / reps, 3)
        print(out, flush=True)
