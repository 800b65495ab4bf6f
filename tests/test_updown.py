"""cs_updown (csparse.py:2318-2365).  The expected factors are what the UNMODIFIED reference's cs_updown made of
the inputs (tests/golden/updown.npz, oracle/gen_golden.py::updown_fixture): update, downdate of the result, and a
downdate that is not positive definite and stops part way.  CPU: the oracle's restatement; GPU: the HIP kernel
through the drop-in module, bit for bit."""
import hashlib

import numpy as np
import pytest

import csparse_oracle as O
from conftest import golden, golden_meta, unpack


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes()).hexdigest()


def _inputs(mod, name):
    """(L, W, W2, parent) as `mod` objects; the factor of bcsstk16 is recomputed (too large to store)."""
    g, m = golden("updown"), golden_meta()["updown"][name]
    n = m["n"]
    pre = name + "_"
    parent = [int(v) for v in g[pre + "parent"]]
    if name == "bcsstk01":
        L = mod.cs_spalloc(n, n, m["lnz"], True, False)
        L.p, L.i, L.x = g[pre + "L_p"].tolist(), g[pre + "L_i"].tolist(), g[pre + "L_x"].tolist()
    else:
        C = unpack(O, golden(name), "C")
        No = O.cs_chol(C, O.cs_schol(0, C))
        L = mod.cs_spalloc(n, n, m["lnz"], True, False)
        L.p, L.i, L.x = list(No.L.p), list(No.L.i), list(No.L.x)
    ws = []
    for key in ("W_x", "W2_x"):
        W = mod.cs_spalloc(n, 1, n, True, False)
        cnt = len(g[pre + "W_i"])
        W.p = [0, cnt]
        W.i = g[pre + "W_i"].tolist() + [0] * (n - cnt)
        W.x = g[pre + key].tolist() + [0.0] * (n - cnt)
        ws.append(W)
    return L, ws[0], ws[1], parent, m, g, pre


def _check(mod, name):
    L, W, W2, parent, m, g, pre = _inputs(mod, name)
    lnz = m["lnz"]
    xlist = L.x
    assert mod.cs_updown(L, +1, W, parent) is m["ok_update"] is True
    assert L.x is xlist                                      # updated in place, like the reference
    assert _sha(L.x[:lnz]) == m["sha_up"]
    if name == "bcsstk01":
        assert np.asarray(L.x[:lnz]).tobytes() == g[pre + "up"].tobytes()
    assert mod.cs_updown(L, -1, W, parent) is True
    assert _sha(L.x[:lnz]) == m["sha_down"]
    assert mod.cs_updown(L, -1, W2, parent) is False         # not positive definite: stops part way ...
    assert _sha(L.x[:lnz]) == m["sha_not_pd"] != m["sha_down"]   # ... having changed L exactly as the reference does
    # bad input / empty vector
    assert mod.cs_updown(None, 1, W, parent) is False and mod.cs_updown(L, 1, W, None) is False
    E = mod.cs_spalloc(m["n"], 1, 1, True, False)
    assert mod.cs_updown(L, 1, E, parent) is True and _sha(L.x[:lnz]) == m["sha_not_pd"]


@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_oracle_updown_matches_reference(name):
    _check(O, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bcsstk01", "bcsstk16"])
def test_device_updown_matches_reference(name):
    import _csx
    import csparse as cs
    _csx.init()
    _check(cs, name)


@pytest.mark.gpu
def test_device_updown_on_a_device_backed_factor_and_solve_after():
    """L from cs_chol stays on the device; after the update the solve with L must equal the oracle's solve with
    the oracle's updated factor (plans cached before the update must not be reused)."""
    import _csx
    import csparse as cs
    import c_oracle as CO
    _csx.init()
    C = unpack(cs, golden("bcsstk01"), "C")
    Co = unpack(O, golden("bcsstk01"), "C")
    n = C.n
    S, So = cs.cs_schol(0, C), O.cs_schol(0, Co)
    N, No = cs.cs_chol(C, S), O.cs_chol(Co, So)
    b = golden("bcsstk01")["b"]
    x0 = b.tolist()
    assert cs.cs_lsolve(N.L, x0)                               # builds (and caches) a plan on the old values
    _, W, _, parent, m, g, pre = _inputs(cs, "bcsstk01")
    _, Wo, _, _, _, _, _ = _inputs(O, "bcsstk01")
    assert cs.cs_updown(N.L, +1, W, S.parent) is True and O.cs_updown(No.L, +1, Wo, So.parent) is True
    x1, xo = b.tolist(), b.tolist()
    assert cs.cs_lsolve(N.L, x1) and cs.cs_ltsolve(N.L, x1)
    O.cs_lsolve(No.L, xo)
    O.cs_ltsolve(No.L, xo)
    # cs_chol's L.x agrees with the oracle's to rounding, so the solves do too: two factors of one matrix (the updated one,
    # L L'), its conditioning in the bound (tests/tol.py)
    import tol as TOL
    nz = No.L.p[n]
    Lm = TOL.csc(n, No.L.p, No.L.i[:nz], No.L.x[:nz])
    assert TOL.normwise(x1, xo) <= TOL.cross_bound(TOL.cond1(Lm @ Lm.T))
