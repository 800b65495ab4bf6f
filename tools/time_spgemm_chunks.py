"""C = S * S' at 1M x 1M / 32 per column for several chunk counts of the one-pass path (spgemm.chunks; 1 = unchunked)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csparse.py_amd"))
import _csx
_csx.init()
lib = _csx.lib()
n, per_col = 1000000, 32
hA, hT = _csx.new_handle(), _csx.new_handle()
_csx.check(lib.csx_gen_grand_uniform(n, per_col, 20240603, hA))
_csx.check(lib.csx_transpose(hA, 1, hT))
for chunks in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8,16,32,64,128").split(",")]:
    _csx.check(lib.csx_set_option(b"spgemm.chunks", chunks))
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
    print("spgemm.chunks=%d: %.2f ms per multiply" % (chunks, best * 1e3), flush=True)
