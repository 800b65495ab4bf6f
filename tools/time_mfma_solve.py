"""The matrix-core solve of config 5 alone (csx_cholsol_factor's plan, rounding-equal order): median and minimum of `sets` timings of
`reps` back-to-back batches, HIP events on the library's stream.  For A/B runs of two builds in one gpurun call (CSX_LIB).
usage: time_mfma_solve.py [nblocks] [bs] [nrhs] [sets] [reps] [exact order 0/1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import _csx
_csx.init(0)
lib = _csx.lib()
C = _csx.C
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 78125
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sets = int(sys.argv[4]) if len(sys.argv) > 4 else 7
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
exact = int(sys.argv[6]) if len(sys.argv) > 6 else 0
n = nb * bs
hA = _csx.new_handle()
_csx.check(lib.csx_gen_gspd(nb, bs, 20240606, hA))
hL, plan = _csx.new_handle(), _csx.new_handle()
_csx.check(lib.csx_cholsol_factor(hA, 0, hL, plan), "cholsol_factor")
_csx.check(lib.csx_cholsol_set_order(plan, exact))
hB = _csx.new_handle()
_csx.check(lib.csx_gen_rhs(n, k, 0, hB))
for _ in range(3):
    _csx.check(lib.csx_cholsol_solve(plan, hB, k))
ms = []
for _ in range(sets):
    with _csx.Timer() as tm:
        for _ in range(reps):
            _csx.check(lib.csx_cholsol_solve(plan, hB, k))
    ms.append(tm.ms / reps)
a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
_csx.check(lib.csx_cholsol_info(plan, a, b, c))
lnz = nb * bs * (bs + 1) // 2
gb = (12.0 * lnz + 4.0 * (n + 1) + 16.0 * n * k) / 1e9
med = sorted(ms)[len(ms) // 2]
print("%d blocks of %d, %d right-hand sides, path %d: median %.3f ms, min %.3f ms (%.2f GB fused count -> %.3f of 8 TB/s at the median)"
      % (nb, bs, k, a.value, med, min(ms), gb, gb / (med / 1e3) / 8000.0), flush=True)
