import sys, time
import numpy as np
sys.path[:0] = ["csparse.py_amd", "oracle", "tests"]
import csparse as cs, _csx
import c_oracle as CO
from conftest import golden, unpack
_csx.init(0)
g = golden("bcsstk16")
A = cs.cs_pin(unpack(cs, g, "C"))
n = A.n
for rep in range(2):
    t = time.perf_counter(); S = cs.cs_schol(0, A); t1 = time.perf_counter() - t
    t = time.perf_counter(); N = cs.cs_chol(A, S); _csx.sync(); t2 = time.perf_counter() - t
    b = cs.dvec(np.ones((n, 64)))
    F = None
    t = time.perf_counter(); ok = cs.cs_cholsol(0, A, b); _csx.sync(); t3 = time.perf_counter() - t
    print("bcsstk16 n=%d lnz=%d: schol %.1f ms, chol %.1f ms, cholsol(64 rhs, incl. both) %.1f ms" % (n, S.lnz, t1*1e3, t2*1e3, t3*1e3), flush=True)
Fac = cs.cholsol_factor(A)
B = cs.dvec(np.ones((n, 64)))
Fac.solve(B); _csx.sync()
with _csx.Timer() as tm:
    for _ in range(5): Fac.solve(B)
print("solve phase 64 rhs: %.3f ms" % (tm.ms / 5), Fac.info())
p = np.asarray(A.p, np.int32); i = np.asarray(A.i[:A.p[n]], np.int32); x = np.asarray(A.x[:A.p[n]])
t = time.perf_counter(); parent, cp = CO.schol(n, p, i); t1 = time.perf_counter() - t
t = time.perf_counter(); L = CO.chol(n, p, i, x, parent, cp); t2 = time.perf_counter() - t
print("C oracle: schol %.1f ms chol %.1f ms" % (t1*1e3, t2*1e3))
