"""Shared test plumbing: markers, paths, golden-fixture loaders."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the product package directory holds the drop-in module `csparse` (+ `_csx`)
for p in (os.path.join(ROOT, "csparse.py_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def unpack(mod, g, prefix):
    """Build a CSC `cs` object of module `mod` (oracle or product) from fixture keys."""
    m, n, nzmax, leni, lenx = (int(v) for v in g[prefix + "_mn"])
    A = mod.cs_spalloc(m, n, max(leni, 1), lenx >= 0, False)
    A.p = [int(v) for v in g[prefix + "_p"]]
    ii = [int(v) for v in g[prefix + "_i"]]
    A.i = ii + [0] * (leni - len(ii))
    if lenx >= 0:
        xx = [float(v) for v in g[prefix + "_x"]]
        A.x = xx + [0.0] * (lenx - len(xx))
    else:
        A.x = None
    A.nzmax = nzmax
    return A


def same_csc(C, g, prefix, exact_x=True, rtol=1e-10):
    """Assert a CSC object equals the fixture: p and i[:nnz] bit-exact, shapes as stored."""
    m, n, nzmax, leni, lenx = (int(v) for v in g[prefix + "_mn"])
    assert (C.m, C.n) == (m, n)
    assert C.nz == -1
    p = [int(v) for v in g[prefix + "_p"]]
    assert list(C.p) == p
    nnz = p[-1]
    assert list(C.i[:nnz]) == [int(v) for v in g[prefix + "_i"]]
    assert C.nzmax == nzmax and len(C.i) == leni
    if lenx < 0:
        assert C.x is None
    else:
        assert len(C.x) == lenx
        ref = g[prefix + "_x"]
        got = np.asarray(C.x[:nnz], dtype=np.float64)
        if exact_x:
            assert got.tobytes() == ref.tobytes()
        else:
            # entries that cancelled to (nearly) nothing carry rounding of their terms
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-12 * float(np.max(np.abs(ref))) if nnz else 0)


@pytest.fixture(scope="session")
def meta():
    return golden_meta()
