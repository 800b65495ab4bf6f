#!/usr/bin/env python3
"""Interleaved A/B timing of k_gaxpy_tiled variants in ONE process (needs build.py --ablation).
code = VARIANT + 100 * waves + 10000 * groups-per-step.   usage: tools/ablate_tiled.py CODE [CODE ...] [--n N] [--gen uniform]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CSX_LIB", os.path.join(ROOT, "csparse.py_amd", "libcsx_ablation.so"))
sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
import _csx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("codes", nargs="+")
ap.add_argument("--n", type=int, default=5000000)
ap.add_argument("--per-col", type=int, default=64)
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--passes", type=int, default=10)
ap.add_argument("--gen", default="strat", choices=["strat", "uniform"])
a = ap.parse_args()
_csx.init(0)
lib = _csx.lib()
hA, hx, hy = _csx.new_handle(), _csx.new_handle(), _csx.new_handle()
gen = lib.csx_gen_grand if a.gen == "strat" else lib.csx_gen_grand_uniform
_csx.check(gen(a.n, a.per_col, 20240602, hA), "gen")
_csx.check(lib.csx_gen_vec(a.n, 7, 0.5, 1.5, hx), "gen_vec")
_csx.check(lib.csx_vec_alloc(a.n, hy), "vec_alloc")
_csx.check(lib.csx_gaxpy_prepare(hA, _csx.GAXPY_TILED), "prepare")
by = 12 * a.n * a.per_col + 4 * (a.n + 1) + 8 * a.n + 16 * a.n
times = {c: [] for c in a.codes}
for rep in range(a.reps + 1):
    for c in a.codes:
        os.environ["CSX_TILED_VARIANT"] = c
        _csx.check(lib.csx_gaxpy(hA, hx, hy, _csx.GAXPY_TILED), "gaxpy")
        with _csx.Timer() as tm:
            for _ in range(a.passes):
                _csx.check(lib.csx_gaxpy(hA, hx, hy, _csx.GAXPY_TILED), "gaxpy")
        if rep:
            times[c].append(tm.ms / a.passes)
for c in a.codes:
    t = times[c]
    med = statistics.median(t)
    print("code %7s  median %.4f ms  min %.4f  max %.4f  frac(median) %.3f" % (c, med, min(t), max(t), by / med / 1e6 / 8000.0))
