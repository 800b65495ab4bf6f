import sys, os, time
sys.path.insert(0, "/root/repo/csparse.py_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import csparse as cs, _csx
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs
lib = _csx.lib(); C = _csx.C
n, Ap, Ai, Ax = _w_matrix(1493)
A = _host_cs(cs, n, n, Ap, Ai, Ax)
F = cs.lusol_factor(A, 0, 1.0, exact=True)
L, U = cs.cs_pin(F.factors.L), cs.cs_pin(F.factors.U)
k = 1024
B = np.random.default_rng(0).uniform(-1, 1, size=(n, k))
for name, M, fn, kind in (("L", L, cs.cs_lsolve, cs.TRI_L), ("U", U, cs.cs_usolve, cs.TRI_U), ("Lt", L, cs.cs_ltsolve, cs.TRI_LT), ("Ut", U, cs.cs_utsolve, cs.TRI_UT)):
    X = cs.dvec(B)
    assert fn(M, X) is True
    plan = M._dev.plans[kind]
    nc = C.c_int32(0)
    _csx.check(lib.csx_tri_components(plan, nc))
    with _csx.Timer() as tm:
        for _ in range(5):
            _csx.check(lib.csx_tri_solve(plan, X.handle, k))
    X64 = cs.dvec(np.ascontiguousarray(B[:, :64]))
    _csx.check(lib.csx_tri_solve(plan, X64.handle, 64))
    with _csx.Timer() as tm64:
        for _ in range(5):
            _csx.check(lib.csx_tri_solve(plan, X64.handle, 64))
    print(name, "64 right-hand sides: %.3f ms" % (tm64.ms / 5))
    nnz = len(M.i) if hasattr(M, "i") else -1
    print(name, "components", nc.value, "of n", n, "-> per block %.2f" % (nc.value / 1493.0), " sweep of 1024: %.3f ms" % (tm.ms / 5), "nnz", M.p[-1])
