"""Config 5's factor pipeline through the C ABI, as bench.py calls it: csx_schol + csx_chol + csx_cholsol_plan on the
5M-row block-SPD matrix (wall times per call; CSX_CHOL_TIMING=1 prints csx_chol's own laps).
usage: time_factor_abi.py [nblocks] [bs] [reps] [clique 0/1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(0)
lib = _csx.lib()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 78125
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
clique = int(sys.argv[4]) if len(sys.argv) > 4 else 1
_csx.check(lib.csx_set_option(b"chol.clique", clique))
n = nb * bs
hB = _csx.new_handle()
_csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hB))
_csx.sync()
for rep in range(reps):
    parent = np.empty(n, dtype=np.int32)
    cp = np.empty(n + 1, dtype=np.int32)
    t0 = time.perf_counter()
    _csx.check(lib.csx_schol(hB, _csx.pi(parent), _csx.pi(cp)), "schol")
    t1 = time.perf_counter()
    hL = _csx.new_handle()
    _csx.check(lib.csx_chol(hB, _csx.pi(parent), _csx.pi(cp), None, hL), "chol")
    _csx.sync()
    t2 = time.perf_counter()
    plan = _csx.new_handle()
    _csx.check(lib.csx_cholsol_plan(hL, None, plan), "plan")
    _csx.sync()
    t3 = time.perf_counter()
    _csx.check(lib.csx_cholsol_set_order(plan, 0), "order")      # the rounding-equal order: matrix-core fragments are built here
    _csx.sync()
    t4 = time.perf_counter()
    print("clique %d  csx_schol %.1f ms  csx_chol %.1f ms  csx_cholsol_plan %.1f ms  total %.1f ms  lnz %d   (+ csx_cholsol_set_order(0) %.1f ms)" %
          (clique, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), int(cp[n]), 1e3 * (t4 - t3)), flush=True)
    _csx.free(plan)
    _csx.free(hL)
