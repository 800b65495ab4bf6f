#!/usr/bin/env python3
"""What a list-based (drop-in) cs_gaxpy call costs on bcsstk16 (290 378 entries), piece by piece: the reference's
own pure-Python loop, list -> numpy conversion, upload + stable transpose + exact kernel + download."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx, csparse as cs, csparse_oracle as O
from conftest import golden, unpack
_csx.init(0); lib = _csx.lib()
g = golden("bcsstk16")
A, Ao = unpack(cs, g, "C"), unpack(O, g, "C")
n = A.n
x = [1.0 + j / n for j in range(n)]
def best(fn, reps=5):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); _csx.sync(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
y = [0.0] * n
out = {"pure_python_port_ms": best(lambda: O.cs_gaxpy(Ao, x, [0.0] * n), 3)}
out["list_call_unpinned_ms"] = best(lambda: cs.cs_gaxpy(A, x, y))
nnz = A.p[n]
out["  of which list->numpy_ms"] = best(lambda: (_csx.i32(A.p[:n + 1]), _csx.i32(A.i[:nnz]), _csx.f64(A.x[:nnz]), _csx.f64(x), _csx.f64(y)))
p_, i_, x_ = _csx.i32(A.p[:n + 1]), _csx.i32(A.i[:nnz]), _csx.f64(A.x[:nnz])
xv, yv = _csx.f64(x), _csx.f64(y)
out["  of which csx_gaxpy_host (upload, transpose, kernel, download)_ms"] = best(lambda: lib.csx_gaxpy_host(n, n, _csx.pi(p_), _csx.pi(i_), _csx.pd(x_), _csx.pd(xv), _csx.pd(yv)))
cs.cs_pin(A)
out["list_call_pinned_ms"] = best(lambda: cs.cs_gaxpy(A, x, y))
dx, dy = cs.dvec(x), cs.dvec(n)
out["device_vectors_pinned_exact_ms"] = best(lambda: cs.cs_gaxpy(A, dx, dy, cs.GAXPY_EXACT))
out["device_vectors_pinned_wave_ms"] = best(lambda: cs.cs_gaxpy(A, dx, dy, cs.GAXPY_WAVE))
print({k: round(v, 3) for k, v in out.items()})
