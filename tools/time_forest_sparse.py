"""cs_schol + cs_chol + solve plan on a forest of small SPARSE trees at scale: block-diagonal, blocks of `bs` columns that are
tridiagonal plus a full last row / column (an arrow): small elimination trees that are not cliques.
usage: time_forest_sparse.py [nblocks] [bs] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import numpy as np
import _csx
_csx.init(0)
lib = _csx.lib()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = nb * bs
# one block's pattern (full symmetric storage, rows ascending), then tiled
cols = []
for c in range(bs):
    rows = {c, bs - 1}
    if c > 0:
        rows.add(c - 1)
    if c + 1 < bs:
        rows.add(c + 1)
    if c == bs - 1:
        rows = set(range(bs))
    cols.append(sorted(rows))
bi = np.concatenate([np.asarray(r, np.int64) for r in cols])
bp = np.concatenate([[0], np.cumsum([len(r) for r in cols])])
bx = np.concatenate([[(8.0 + (c % 5)) if r == c else -1.0 / (1 + abs(r - c)) for r in cols[c]] for c in range(bs)])
Ap = (np.arange(nb)[:, None] * bp[-1] + bp[None, :-1]).reshape(-1)
Ap = np.concatenate([Ap, [nb * bp[-1]]]).astype(np.int32)
Ai = (bi[None, :] + (np.arange(nb) * bs)[:, None]).reshape(-1).astype(np.int32)
Ax = np.tile(bx, nb)
hA = _csx.new_handle()
_csx.check(lib.csx_csc_upload(n, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
print("n %d nnz %d blocks %d of %d" % (n, len(Ai), nb, bs), flush=True)
for rep in range(reps):          # round 5: the same as ONE call (csx_cholsol_factor, rounding-equal order: block list + fragments from L's columns)
    hLf, planf = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_csc_invalidate(hA))
    _csx.sync()
    t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_factor(hA, 0, hLf, planf), "cholsol_factor")
    _csx.sync()
    print("csx_cholsol_factor (one call) %.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
    _csx.free(planf)
    _csx.free(hLf)
_csx.check(lib.csx_csc_invalidate(hA))
for rep in range(reps):
    parent, cp = np.empty(n, np.int32), np.empty(n + 1, np.int32)
    t0 = time.perf_counter()
    _csx.check(lib.csx_schol(hA, _csx.pi(parent), _csx.pi(cp)), "schol")
    t1 = time.perf_counter()
    hL = _csx.new_handle()
    _csx.check(lib.csx_chol(hA, _csx.pi(parent), _csx.pi(cp), None, hL), "chol")
    _csx.sync()
    t2 = time.perf_counter()
    plan = _csx.new_handle()
    _csx.check(lib.csx_cholsol_plan(hL, None, plan), "plan")
    _csx.sync()
    t3 = time.perf_counter()
    print("csx_schol %.1f ms  csx_chol %.1f ms  csx_cholsol_plan %.1f ms  lnz %d" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), int(cp[n])), flush=True)
    if rep == reps - 1:      # the solve phase on that plan: 128 right-hand sides, both orders
        k = 128
        for exact in (1, 0):
            _csx.check(lib.csx_cholsol_set_order(plan, exact))
            hB = _csx.new_handle()
            _csx.check(lib.csx_gen_rhs(n, k, 0, hB))
            _csx.check(lib.csx_cholsol_solve(plan, hB, k))
            with _csx.Timer() as tm:
                for _ in range(5):
                    _csx.check(lib.csx_cholsol_solve(plan, hB, k))
            a, b, c = _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32()
            _csx.check(lib.csx_cholsol_info(plan, a, b, c))
            gb = (12.0 * int(cp[n]) * 2 + 16.0 * n * k) / 1e9
            print("cholsol solve, %d right-hand sides, exact=%d: %.3f ms (path %d, %d trees, widest %d; %.2f GB fused count -> %.0f GB/s)"
                  % (k, exact, tm.ms / 5, a.value, b.value, c.value, gb, gb / (tm.ms / 5 / 1e3)), flush=True)
            _csx.free(hB)
    _csx.free(plan)
    _csx.free(hL)
