"""cs_schol + cs_chol of the 5M-row block-SPD matrix of the cholsol leg (for a kernel trace / wall times)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import _csx, csparse as cs
_csx.init(0)
lib = _csx.lib()
nb, bs = 78125, 64
hB = _csx.new_handle()
_csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hB))
A = cs._from_device(hB, lambda nnz: max(nnz, 1))
A._pinned = True
for rep in range(2):
    _csx.sync(); t0 = time.perf_counter()
    S = cs.cs_schol(0, A)
    _csx.sync(); t1 = time.perf_counter()
    N = cs.cs_chol(A, S)
    _csx.sync(); t2 = time.perf_counter()
    print("cs_schol %.3f s  cs_chol %.3f s  lnz %d" % (t1 - t0, t2 - t1, S.lnz))
