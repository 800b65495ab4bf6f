"""cs_lu by the column elimination tree on the device (csx_lu_etree) against the host loop (csx_lu_host, one core) on the
shape the planner takes by itself: bordered block matrices (shallow bushy tree, short reaches; tests/test_gpu_lu_etree.py
_bordered_blocks).  usage: time_lu_bordered.py [blocks ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import _csx
_csx.init(0)
_csx.check(_csx.lib().csx_set_option(b"lu.etree", 1))     # the planner's own criteria (the default since round 4 is 0: never)
from test_gpu_lu_etree import _bordered_blocks, _device_lu, _host_lu
for nb in [int(v) for v in sys.argv[1:]] or [40, 400, 4000]:
    n, Ap, Ai, Ax = _bordered_blocks(nb)
    _device_lu(Ap, Ai, Ax, 1.0)
    t0 = time.perf_counter(); dev = _device_lu(Ap, Ai, Ax, 1.0); _csx.sync(); td = time.perf_counter() - t0
    t0 = time.perf_counter(); host = _host_lu(n, Ap, Ai, Ax, 1.0); th = time.perf_counter() - t0
    same = dev != "host" and all(a.tobytes() == b.tobytes() for a, b in zip(dev, host))
    print({"blocks": nb, "n": n, "nnz": int(Ap[-1]), "device_took_it": dev != "host", "device_s_incl_upload_download": round(td, 4),
           "host_one_core_s": round(th, 4), "bit_identical": same}, flush=True)
