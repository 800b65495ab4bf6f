"""cs_chol on chain-like banded factors: bcsstk16 (half-width 140) and 2-D grid Laplacians in natural order
(half-width g), register-window kernel against the blocked dense-band kernels, bits against the plain-C oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import scipy.sparse as sp
import _csx, csparse as cs
import c_oracle as CO
_csx.init(0)


def grid(g):
    n = g * g
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(g, g))
    A = (sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)) + 0.01 * sp.identity(n)).tocsc()
    A.sort_indices()
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def run(name, n, p, i, x, check=True):
    M = cs.cs_spalloc(n, n, len(i), True, False)
    M.p, M.i, M.x = p.tolist(), i.tolist(), x.tolist()
    cs.cs_pin(M)
    S = cs.cs_schol(0, M)
    ref = None
    if check:
        parent, cp = CO.schol(n, p, i)
        t0 = time.perf_counter()
        Lp, Li, Lx = CO.chol(n, p, i, x, parent, cp)
        t_c = time.perf_counter() - t0
        ref = Lx
    else:
        t_c = None
    for wb, nb in ((2, 16), (2, 32), (2, -16), (0, 16)):
        if wb == 0 and S.lnz > 3e7:
            continue
        with _csx.option("chol.wband", wb), _csx.option("chol.wband_nb", nb):
            N = cs.cs_chol(M, S); _csx.sync()
            t0 = time.perf_counter(); N = cs.cs_chol(M, S); _csx.sync(); dt = time.perf_counter() - t0
        out = {"matrix": name, "n": n, "lnz": int(S.lnz), "wband": wb, "nb": nb, "chol_ms": round(dt * 1e3, 3),
               "host_core_ms": None if t_c is None else round(t_c * 1e3, 1)}
        if ref is not None:
            got = np.asarray(cs.cs_host(N.L).x[:len(ref)]) if hasattr(cs, "cs_host") else np.asarray(N.L.x[:len(ref)])
            out["bit_identical"] = got.tobytes() == ref.tobytes()
            out["max_rel"] = float(np.max(np.abs(got - ref)) / np.abs(ref).max())
        print(out, flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "bcsstk16"
if which == "bcsstk16":
    g = np.load(os.path.join(ROOT, "tests", "golden", "bcsstk16.npz"))
    run("bcsstk16", int(g["C_p"].shape[0] - 1), g["C_p"].astype(np.int32), g["C_i"].astype(np.int32), g["C_x"])
else:
    gsz = int(which)
    n, p, i, x = grid(gsz)
    run("grid%d" % gsz, n, p, i, x, check=(len(sys.argv) < 3 or sys.argv[2] != "nocheck"))
