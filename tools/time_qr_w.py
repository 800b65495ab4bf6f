"""cs_qr on W (BASELINE config 3's matrix, 1 493 blocks of west0067): device block path against the host C++ code."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import _csx, csparse as cs
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs
_csx.init(0)
n, Ap, Ai, Ax = _w_matrix(1493)
A = cs.cs_pin(_host_cs(cs, n, n, Ap, Ai, Ax))
t0 = time.perf_counter(); S = cs.cs_sqr(0, A, True); t_sqr = time.perf_counter() - t0
N = cs.cs_qr(A, S); _csx.sync()
t0 = time.perf_counter(); N = cs.cs_qr(A, S); _csx.sync(); t_dev = time.perf_counter() - t0
assert N.L._lazy
parent, pinv, leftmost = _csx.i32(S.parent), _csx.i32(S.pinv), _csx.i32(S.leftmost)
vcap, rcap = max(int(S.lnz), 1), max(int(S.unz), 1)
Vp, Rp = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
Vi, Ri = np.zeros(vcap, np.int32), np.zeros(rcap, np.int32)
Vx, Rx, beta = np.zeros(vcap), np.zeros(rcap), np.zeros(n)
t0 = time.perf_counter()
st = _csx.load().csx_qr_host(n, n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), None, _csx.pi(parent), _csx.pi(pinv),
                             _csx.pi(leftmost), vcap, rcap, _csx.pi(Vp), _csx.pi(Vi), _csx.pd(Vx), _csx.pi(Rp),
                             _csx.pi(Ri), _csx.pd(Rx), _csx.pd(beta))
t_host = time.perf_counter() - t0
print({"n": n, "nnz_V": int(Vp[n]), "nnz_R": int(Rp[n]), "cs_sqr_s": round(t_sqr, 3), "cs_qr_device_ms": round(t_dev * 1e3, 2),
       "cs_qr_host_cpp_ms": round(t_host * 1e3, 2), "same_beta": bool(np.asarray(N.B).tobytes() == beta.tobytes())})
F = cs.qrsol_factor(A)
B = np.repeat((1.0 + np.arange(n) / n)[:, None], 64, axis=1)
X = F.solve(cs.dvec(B)); _csx.sync()
t0 = time.perf_counter(); X = F.solve(cs.dvec(B)); _csx.sync(); t_solve = time.perf_counter() - t0
Xq = cs.dvec(n, 64)
with _csx.Timer() as tm:
    cs.apply_q(F.factors, Xq, True)
print({"qrsol_solve_64rhs_ms": round(t_solve * 1e3, 2), "apply_q_64rhs_ms": round(tm.ms, 2)})
