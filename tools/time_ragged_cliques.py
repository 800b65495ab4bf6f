"""cs_cholsol on a forest of dense cliques of UNEQUAL sizes at scale (north_star's "batches of independent matrices" are unequal
in general): block-diagonal SPD, block sizes drawn uniformly from [lo, hi], about n rows.  csx_cholsol_factor (one call), then
the solve of 128 right-hand sides in the exact order (fused per-tree kernel) and in the rounding-equal order (csx_trimfma.hip:
trees made dense by size class on the matrix cores), each column of a sample against the other order.
usage: time_ragged_cliques.py [n] [lo] [hi] [nrhs] [chol.exact 1/0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(0)
lib = _csx.lib()
C = _csx.C
n_want = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = int(sys.argv[4]) if len(sys.argv) > 4 else 128
_csx.check(lib.csx_set_option(b"chol.exact", int(sys.argv[5]) if len(sys.argv) > 5 else 1))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
t0 = time.perf_counter()
n, Ap32, Ai, Ax, sizes = synth.ragged_cliques(n_want, lo, hi, 20240605)
nnz = int(Ap32[-1])
print("n %d nnz %d blocks %d of %d..%d columns (made in %.1f s)" % (n, nnz, len(sizes), lo, hi, time.perf_counter() - t0), flush=True)
hA = _csx.new_handle()
_csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap32), _csx.pi(Ai), _csx.pd(Ax), hA))
del Ai, Ax
lnz = int(np.sum(sizes.astype(np.int64) * (sizes + 1) // 2))
for rep in range(3):
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.sync()
    t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan), "cholsol_factor")
    _csx.sync()
    dt = time.perf_counter() - t0
    path, fa, fn, fc = C.c_int32(-1), C.c_double(0), C.c_double(0), C.c_double(0)
    _csx.check(lib.csx_cholsol_factor_info(path, fa, fn, fc))
    print("csx_cholsol_factor %.2f ms (path %d, analysis %.2f ms, block kernel %.3f ms)  lnz %d" % (dt * 1e3, path.value, fa.value, fn.value, lnz), flush=True)
    if rep < 2:
        _csx.free(plan)
        _csx.free(hL)
gb = (12.0 * lnz + 4.0 * (n + 1) + 16.0 * n * k) / 1e9         # the fused count: L once, B read once, X written once
sols = {}
for exact in (1, 0):
    _csx.check(lib.csx_cholsol_set_order(plan, exact))
    hB = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
    _csx.check(lib.csx_cholsol_solve(plan, hB, k))
    x = np.empty(n * k)
    _csx.check(lib.csx_vec_download(hB, _csx.pd(x), n * k))
    sols[exact] = x.reshape(n, k)[:, [0, k // 2, k - 1]].copy()
    for _ in range(2):
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
    sets = []
    for _ in range(5):
        with _csx.Timer() as tm:
            for _ in range(5):
                _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        sets.append(tm.ms / 5)
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _csx.check(lib.csx_cholsol_info(plan, a, b, c))
    g = C.c_double(0)
    _csx.check(lib.csx_cholsol_growth(plan, g))
    ms = sorted(sets)[2]                                      # median of five sets of five
    print("cholsol solve, %d right-hand sides, exact=%d: %.3f ms (path %d, %d trees, widest %d, guard %.1f; %.2f GB fused count -> %.0f GB/s = %.3f of 8 TB/s)"
          % (k, exact, ms, a.value, b.value, c.value, g.value, gb, gb / (ms / 1e3), gb / (ms / 1e3) / 8000.0), flush=True)
    _csx.free(hB)
err = np.max(np.abs(sols[0] - sols[1]) / np.abs(sols[1]))
print("rounding-equal against exact, three columns: max componentwise relative difference %.2e" % err, flush=True)
