"""Config 5's factor pipeline as ONE library call (round 5): csx_cholsol_factor = cs_schol + cs_chol + the solve plan with the
symbolic analysis staying on the device, on the 5M-row block-SPD matrix; then one batch of right-hand sides on that plan.
usage: time_factor_fused.py [nblocks] [bs] [reps] [chol.exact 1/0] [nrhs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(0)
lib = _csx.lib()
C = _csx.C
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 78125
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
exact_chol = int(sys.argv[4]) if len(sys.argv) > 4 else 1
k = int(sys.argv[5]) if len(sys.argv) > 5 else 128
_csx.check(lib.csx_set_option(b"chol.exact", exact_chol))
n = nb * bs
hA = _csx.new_handle()
_csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
_csx.sync()
for rep in range(reps):
    hL, plan = _csx.new_handle(), _csx.new_handle()
    _csx.sync()
    t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan), "cholsol_factor")
    _csx.sync()
    dt = time.perf_counter() - t0
    path, fa, fn, fc = C.c_int32(-1), C.c_double(0), C.c_double(0), C.c_double(0)
    _csx.check(lib.csx_cholsol_factor_info(path, fa, fn, fc))
    print("chol.exact %d  csx_cholsol_factor %.3f ms (path %d: analysis %.3f ms, block kernel %.4f ms by HIP events)" %
          (exact_chol, dt * 1e3, path.value, fa.value, fn.value), flush=True)
    if rep == reps - 1:
        hB = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        with _csx.Timer() as tm:
            for _ in range(5):
                _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _csx.check(lib.csx_cholsol_info(plan, a, b, c))
        print("one batch of %d right-hand sides on that plan: %.3f ms (path %d)" % (k, tm.ms / 5, a.value), flush=True)
        _csx.free(hB)
    _csx.free(plan)
    _csx.free(hL)
