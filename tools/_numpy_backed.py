"""cs objects whose p / i / x are numpy arrays instead of lists (the reference only ever indexes them): what an unpinned
cs_gaxpy call costs then, against the list-backed call (tools/time_listcall.py) -- bcsstk16, 290 378 entries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
g = golden("bcsstk16")
A = unpack(cs, g, "C")                      # list-backed
n = A.n
x = np.linspace(0.5, 1.5, n)
def timeit(fn, reps=20):
    fn(); _csx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    _csx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
yl = [0.0] * n
t_list = timeit(lambda: cs.cs_gaxpy(A, x.tolist(), yl))
An = cs.cs_spalloc(n, n, len(A.i), True, False)
An.p, An.i, An.x = np.asarray(A.p, np.int32), np.asarray(A.i, np.int32), np.asarray(A.x, np.float64)
yn = np.zeros(n)
ok = cs.cs_gaxpy(An, x, yn)
ref = [0.0] * n
cs.cs_gaxpy(A, x.tolist(), ref)
print("numpy-backed call works:", ok, "same bits as the list call:", np.asarray(ref).tobytes() == yn.tobytes())
yn[:] = 0
t_np = timeit(lambda: cs.cs_gaxpy(An, x, yn))
print("unpinned cs_gaxpy, bcsstk16: list-backed %.2f ms, numpy-backed %.2f ms per call" % (t_list, t_np))
