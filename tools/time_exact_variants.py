"""Exact-order dense-block cholsol (the default of every plan) on G-spd, 5M rows, 128 right-hand sides: the four kernel
variants ("cholsol.exact_variant": 1 = one fence per row / one RHS per lane, 2 = ring / one, 3 = rows / two, 4 = ring / two, 5 = L values by DPP row broadcast, one term in four by LDS broadcast / one, 6 = DPP only / one)
per block size, ms per batch, and that every variant gives the same bits."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(); lib = _csx.lib()
for bs in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,32,16,8").split(",")]:
    nb = 5000000 // bs; n = nb * bs; k = 128
    hA = _csx.new_handle(); _csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
    parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
    _csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)))
    hL = _csx.new_handle(); _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL))
    plan = _csx.new_handle(); _csx.check(lib.csx_cholsol_plan(hL, None, plan))
    ref = None
    for variant in (1, 2, 3, 4, 5, 6, 5, 6):     # 5 and 6 twice: their difference is inside the run-to-run noise of one pass
        if bs == 64 and variant in (3, 4):
            continue
        _csx.check(lib.csx_set_option(b"cholsol.exact_variant", variant))
        hB = _csx.new_handle(); _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
        _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        head = np.empty(4096 * k)
        _csx.check(lib.csx_vec_download(hB, _csx.pd(head), head.size))
        if ref is None:
            ref = head.tobytes()
        same = head.tobytes() == ref
        with _csx.Timer() as tm:
            for _ in range(3):
                _csx.check(lib.csx_cholsol_solve(plan, hB, k))
        print("bs %d variant %d: %.3f ms per 128 RHS, same bits as variant 1: %s" % (bs, variant, tm.ms / 3, same), flush=True)
        _csx.free(hB)
    _csx.check(lib.csx_set_option(b"cholsol.exact_variant", 0))
    for h in (plan, hL, hA):
        _csx.free(h)
