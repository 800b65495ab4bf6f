"""Exact-order dense-block cholsol (the default of every plan), blocks of 32 and 16: one against two right-hand sides per lane."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(); lib = _csx.lib()
for bs in (32, 16):
    nb = 5000000 // bs; n = nb * bs; k = 128
    hA = _csx.new_handle(); _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
    parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
    _csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)))
    hL = _csx.new_handle(); _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
    plan = _csx.new_handle(); _csx.check(lib.csx_cholsol_plan(hL, None, plan))
    hB = _csx.new_handle(); _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
    for pairs in (0, 1):
        _csx.check(lib.csx_set_option(b"cholsol.exact_pairs", pairs))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        with _csx.Timer() as tm:
            for _ in range(5):
                _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        print("bs %d pairs %d: %.3f ms per 128 RHS" % (bs, pairs, tm.ms / 5), flush=True)
    for h in (plan, hL, hA, hB):
        _csx.free(h)
