"""cs_lsolve / cs_ltsolve / cholsol on the factor of a natural-order grid Laplacian (chain tree, band = grid side):
time per solve, bits against the plain-C oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import scipy.sparse as sp
import _csx, csparse as cs
import c_oracle as CO
_csx.init(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 200
check = len(sys.argv) < 3 or sys.argv[2] != "nocheck"
n = g * g
T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc()
A.sort_indices()
p, i, x = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
M = cs.cs_spalloc(n, n, len(i), True, False)
M.p, M.i, M.x = p.tolist(), i.tolist(), x.tolist()
cs.cs_pin(M)
t0 = time.perf_counter(); S = cs.cs_schol(0, M); N = cs.cs_chol(M, S); _csx.sync(); t_f = time.perf_counter() - t0
b = np.linspace(1.0, 2.0, n)
if check:
    parent, cp = CO.schol(n, p, i)
    Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
    t0 = time.perf_counter(); y_ref = CO.lsolve(n, Lp, Li, Lx, b); t_cl = time.perf_counter() - t0
    t0 = time.perf_counter(); x_ref = CO.ltsolve(n, Lp, Li, Lx, y_ref); t_clt = time.perf_counter() - t0
for k in (1, 16):
    B = np.repeat(b[:, None], k, axis=1) if k > 1 else b.copy()
    d = cs.dvec(B)
    cs.cs_lsolve(N.L, d); _csx.sync()
    d = cs.dvec(B)
    t0 = time.perf_counter(); cs.cs_lsolve(N.L, d); _csx.sync(); t_l = time.perf_counter() - t0
    y = d.numpy().reshape(n, -1)[:, 0].copy()
    t0 = time.perf_counter(); cs.cs_ltsolve(N.L, d); _csx.sync(); t_lt0 = time.perf_counter() - t0
    d2 = cs.dvec(np.repeat(y[:, None], k, axis=1) if k > 1 else y.copy())
    t0 = time.perf_counter(); cs.cs_ltsolve(N.L, d2); _csx.sync(); t_lt = time.perf_counter() - t0
    xs = d2.numpy().reshape(n, -1)[:, -1].copy()
    out = {"grid": g, "n": n, "lnz": int(S.lnz), "nrhs": k, "factor_s": round(t_f, 3), "lsolve_ms": round(t_l * 1e3, 2),
           "ltsolve_ms": round(t_lt * 1e3, 2), "ltsolve_first_ms": round(t_lt0 * 1e3, 2)}
    if check:
        out.update({"lsolve_bits": y.tobytes() == y_ref.tobytes(), "ltsolve_bits": xs.tobytes() == x_ref.tobytes(),
                    "host_core_lsolve_ms": round(t_cl * 1e3, 2), "host_core_ltsolve_ms": round(t_clt * 1e3, 2)})
    print(out, flush=True)
for exact in (True, False):
    F = cs.cholsol_factor(M, 0, exact=exact)
    for k in (1, 16):
        B = cs.dvec(np.repeat(b[:, None], k, axis=1) if k > 1 else b.copy())
        F.solve(B); _csx.sync()
        B = cs.dvec(np.repeat(b[:, None], k, axis=1) if k > 1 else b.copy())
        t0 = time.perf_counter(); F.solve(B); _csx.sync(); dt = time.perf_counter() - t0
        xv = B.numpy().reshape(n, -1)[:, -1]
        res = float(np.max(np.abs(A @ xv - b)))
        out = {"cholsol exact": exact, "nrhs": k, "solve_ms": round(dt * 1e3, 2), "residual_inf": res}
        if check:
            out["bits"] = xv.tobytes() == x_ref.tobytes()
            out["max_rel"] = float(np.max(np.abs(xv - x_ref)) / np.max(np.abs(x_ref)))
        print(out, flush=True)
