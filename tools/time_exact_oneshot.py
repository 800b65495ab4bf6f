import os, sys, time
sys.path.insert(0, "/root/repo/csparse.py_amd")
import _csx
_csx.init(0)
lib = _csx.lib(); C = _csx.C
nb, bs, k = 78125, 64, 128
n = nb * bs
hA = _csx.new_handle()
_csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
for rep in range(4):
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.sync(); t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_factor(hA, 1, hL, plan))
    _csx.sync(); t1 = time.perf_counter()
    hB = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
    _csx.sync(); t2 = time.perf_counter()
    _csx.check(lib.csx_cholsol_solve(plan, hB, k))
    _csx.sync(); t3 = time.perf_counter()
    print("exact plan: csx_cholsol_factor %.2f ms, first solve of 128 %.2f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3), flush=True)
    _csx.free(hB); _csx.free(plan); _csx.free(hL)
