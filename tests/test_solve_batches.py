"""Blocks of right-hand sides against the UNMODIFIED reference, column by column (tests/golden/solve_batches.npz, made by
oracle/gen_golden.py batch: cs_lusol(0, C, b, tol) and cs_qrsol(0, C, b) of csparse.py:1456-1478, :1875-1912 on the square
problem matrices of csparse_test.py).  CPU: the oracle's restatements reproduce the reference's columns (LU with the
shipped loop bit for bit).  GPU: the batched solvers lusol_factor / qrsol_factor -- factor once, every column on the device
-- against the same columns within BASELINE's 1e-10 (the product's LU walks xi[top..n-1], SURVEY D7: other factors, the
same solution to rounding)."""
import numpy as np
import pytest

import csparse_oracle as O
import tol as TOL
from conftest import golden, unpack

NAMES = ["t1", "bcsstk01", "west0067", "fs_183_1"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_references_columns(name, meta):
    g, gb = golden(name), golden("solve_batches")
    B, XL, XQ = gb[name + "_B"], gb[name + "_x_lusol"], gb[name + "_x_qrsol"]
    tol = meta["solve_batches"][name]["tol"]
    for r in range(B.shape[1]):
        v = B[:, r].tolist()
        assert O.cs_lusol(0, unpack(O, g, "C"), v, tol, shipped_quirk=True)
        assert np.asarray(v).tobytes() == np.ascontiguousarray(XL[:, r]).tobytes()
        v = B[:, r].tolist()
        assert O.cs_lusol(0, unpack(O, g, "C"), v, tol)                 # the bounded loop: the same solution to rounding
        assert TOL.normwise(v, XL[:, r]) < TOL.X_RTOL
        # (cs_qrsol has no restatement under oracle/: the product's QR is pinned by the reference's outputs directly -- below,
        # tests/test_gpu_qrsol.py, tests/test_host_symbolic.py); the two reference solvers agree with each other to the
        # conditioning of the matrix
        assert TOL.normwise(XQ[:, r], XL[:, r]) < TOL.cross_bound(TOL.cond1(TOL.csc(B.shape[0], g["C_p"], g["C_i"], g["C_x"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_batched_solvers_against_the_references_columns(name, meta):
    import _csx
    import csparse as cs
    _csx.init(0)
    g, gb = golden(name), golden("solve_batches")
    B, XL, XQ = gb[name + "_B"], gb[name + "_x_lusol"], gb[name + "_x_qrsol"]
    n, k = B.shape
    tol = meta["solve_batches"][name]["tol"]
    C = cs.cs_pin(unpack(cs, g, "C"))
    FL = cs.lusol_factor(C, 0, tol)
    dB = cs.dvec(np.ascontiguousarray(B))
    assert FL.solve(dB) is True
    X = dB.numpy().reshape(n, k)
    for r in range(k):
        assert TOL.normwise(X[:, r], XL[:, r]) < TOL.X_RTOL, (name, r)
        assert TOL.componentwise(X[:, r], XL[:, r]) < TOL.X_RTOL, (name, r)
    FQ = cs.qrsol_factor(C, 0)
    XQd = FQ.solve(cs.dvec(np.ascontiguousarray(B))).numpy().reshape(n, k)
    for r in range(k):
        assert TOL.normwise(XQd[:, r], XQ[:, r]) < TOL.X_RTOL, (name, r)
