import sys, time
sys.path.insert(0, "/root/repo/csparse.py_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, _csx, synth
_csx.init(0); lib = _csx.lib(); C = _csx.C
n, Ap32, Ai, Ax, sizes = synth.ragged_cliques(5000000, 8, 64, 20240605)
hA = _csx.new_handle()
_csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap32), _csx.pi(Ai), _csx.pd(Ax), hA))
for rep in range(5):
    _csx.check(lib.csx_csc_invalidate(hA))
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.sync(); t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan))
    _csx.sync(); dt = time.perf_counter() - t0
    path, fa, fn, fc = C.c_int32(-1), C.c_double(0), C.c_double(0), C.c_double(0)
    _csx.check(lib.csx_cholsol_factor_info(path, fa, fn, fc))
    print("uncached call %.2f ms (analysis %.2f, block kernel %.3f)" % (dt * 1e3, fa.value, fn.value), flush=True)
    _csx.free(plan); _csx.free(hL)
