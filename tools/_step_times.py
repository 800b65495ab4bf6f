"""Per-step times of the tiled cs_gaxpy on G-rand right after other work (a transpose): does the kernel speed up as it keeps running?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import _csx, csparse as cs
_csx.init(0)
lib = _csx.lib(); C = _csx.C
n, per = 5000000, 64
hA, hx, hy = _csx.new_handle(), _csx.new_handle(), _csx.new_handle()
_csx.check(lib.csx_gen_grand_uniform(n, per, 20240602, hA)); _csx.check(lib.csx_gen_vec(n, 7, 0.5, 1.5, hx)); _csx.check(lib.csx_vec_alloc(n, hy))
_csx.check(lib.csx_gaxpy_prepare(hA, cs.GAXPY_TILED)); _csx.sync()
def steps(k):
    out = []
    for _ in range(k):
        _csx.check(lib.csx_timer_start()); _csx.check(lib.csx_gaxpy(hA, hx, hy, cs.GAXPY_TILED))
        ms = C.c_double(0.0); _csx.check(lib.csx_timer_stop(ms)); out.append(round(ms.value, 3))
    return out
for what in ("first", "after transpose", "after 2 s idle", "after 100 more"):
    if what == "after transpose":
        hT = _csx.new_handle(); _csx.check(lib.csx_transpose(hA, 1, hT)); _csx.sync(); _csx.free(hT)
    if what == "after 2 s idle":
        time.sleep(2.0)
    if what == "after 100 more":
        for _ in range(100): _csx.check(lib.csx_gaxpy(hA, hx, hy, cs.GAXPY_TILED))
    print(what, steps(24), flush=True)
