"""cs_chol on bcsstk16, repeated (for a kernel trace): tools/kernel_times.sh tools/time_chol16.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import _csx, csparse as cs
from conftest import golden, unpack
_csx.init(0)
g = golden("bcsstk16")
C = cs.cs_pin(unpack(cs, g, "C"))
S = cs.cs_schol(0, C)
for rep in range(5):
    _csx.sync(); t0 = time.perf_counter(); N = cs.cs_chol(C, S); _csx.sync(); dt = time.perf_counter() - t0
print("cs_chol %.2f ms" % (dt * 1e3))
