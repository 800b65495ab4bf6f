"""Timing only (no checks): C = S * S' at 1M x 1M / 32 per column, ms per multiply.  With CSX_LIB pointing at the ablation
build and CSX_SG_ABL set, parts of the one-pass hash kernel are switched off (results are then wrong by design)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csparse.py_amd"))
import _csx
_csx.init()
lib = _csx.lib()
n, per_col = 1000000, 32
hA, hT = _csx.new_handle(), _csx.new_handle()
_csx.check(lib.csx_gen_grand_uniform(n, per_col, 20240603, hA))
_csx.check(lib.csx_transpose(hA, 1, hT))
best = None
for rep in range(4):
    hC = _csx.new_handle()
    _csx.sync()
    t0 = time.perf_counter()
    _csx.check(lib.csx_multiply(hA, hT, hC))
    _csx.sync()
    dt = time.perf_counter() - t0
    _csx.free(hC)
    if rep:
        best = dt if best is None else min(best, dt)
print("CSX_SG_ABL=%s: %.2f ms per multiply" % (os.environ.get("CSX_SG_ABL", "0"), best * 1e3))
