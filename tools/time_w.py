#!/usr/bin/env python3
"""W (BASELINE config 3) triangular solves through the C ABI in a tight loop: HIP-event time per L+U pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx, csparse as cs
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs
_csx.init(0); lib = _csx.lib()
n, Ap, Ai, Ax = _w_matrix(1493)
A = _host_cs(cs, n, n, Ap, Ai, Ax)
N = cs.cs_lu(A, cs.cs_sqr(0, A, False), 1.0)
L, U = cs.cs_pin(N.L), cs.cs_pin(N.U)
for k in (1, 8, 64, 256, 1024):
    X = cs.dvec(np.ones((n, k)) if k > 1 else np.ones(n))
    cs.cs_lsolve(L, X); cs.cs_usolve(U, X)
    pl, pu = L._dev.plans[cs.TRI_L], U._dev.plans[cs.TRI_U]
    reps = 200
    for _ in range(3):
        lib.csx_tri_solve(pl, X.handle, k); lib.csx_tri_solve(pu, X.handle, k)
    with _csx.Timer() as tm:
        for _ in range(reps):
            lib.csx_tri_solve(pl, X.handle, k); lib.csx_tri_solve(pu, X.handle, k)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.csx_tri_solve(pl, X.handle, k); lib.csx_tri_solve(pu, X.handle, k)
    _csx.sync()
    wall = (time.perf_counter() - t0) / reps
    print("nrhs %4d: %.1f us per L+U pair (HIP events), %.1f us wall" % (k, tm.ms / reps * 1e3, wall * 1e6))
