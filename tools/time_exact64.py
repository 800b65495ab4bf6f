"""Exact-order dense-block cholsol on G-spd (5M x 5M, blocks of 64), 128 right-hand sides: ms per batch; the result against
the rounding-equal kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(); lib = _csx.lib()
bs = 64; nb = 5000000 // bs; n = nb * bs; k = 128
hA = _csx.new_handle(); _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
_csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)))
hL = _csx.new_handle(); _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
plan = _csx.new_handle(); _csx.check(lib.csx_cholsol_plan(hL, None, plan))
hB = _csx.new_handle(); _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
_csx.check(lib.csx_cholsol_solve(plan, hB, k))
with _csx.Timer() as tm:
    for _ in range(5):
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
print("exact order, blocks of 64, 128 RHS: %.3f ms per batch" % (tm.ms / 5), flush=True)
