""""tri.host_chains" (round 5, opt-in, default 0): ONE host right-hand side on a chain-like factor is solved by the reference's loop
on the host inside libcsx (csx_tri_solve_list / csx_cholsol_solve_list; csparse.py:1330-1365, :2368-2385, :2460-2475, :640-643) --
a dispatch for a case the device loses (bcsstk16: 4 810 levels for 4 884 unknowns), same operations, same order, same bits.  Off by
default: every other test of this suite runs the HIP path."""
import ctypes as C

import numpy as np
import pytest

import c_oracle as CO
from conftest import golden, unpack
from test_gpu_parity import _host_cs, cs  # noqa: F401
from test_gpu_tricomponents import _block_tri

pytestmark = pytest.mark.gpu


@pytest.fixture
def host_chains(cs):
    cs.cs_option("tri.host_chains", 1)
    yield
    cs.cs_option("tri.host_chains", 0)


def _taken(cs, plan, n, x):
    import _csx
    buf = np.asarray(x, np.float64).copy()
    t = C.c_int(-1)
    _csx.check(_csx.lib().csx_tri_solve_list(plan, _csx.pd(buf), t))
    return t.value, buf


def test_default_is_off(cs):
    import _csx
    v = C.c_int(-1)
    _csx.check(_csx.lib().csx_get_option(b"tri.host_chains", v))
    assert v.value == 0
    g = golden("bcsstk16")
    L = cs.cs_pin(unpack(cs, g, "Lo"))
    x = g["b"].tolist()
    assert cs.cs_lsolve(L, x) is True                       # the device, as in every other test
    assert np.asarray(x).tobytes() == g["x_lsolve"].tobytes()
    t, buf = _taken(cs, L._dev.plans[cs.TRI_L], L.n, g["b"])
    assert t == 0 and buf.tobytes() == g["b"].tobytes()     # refused, untouched


def test_chain_factor_list_solves_on_the_host_with_the_reference_bits(cs, host_chains):
    g = golden("bcsstk16")
    Cm = unpack(cs, g, "C")
    n = Cm.n
    Cp, Ci, Cx = (np.asarray(v) for v in (Cm.p, Cm.i[:Cm.p[n]], Cm.x[:Cm.p[n]]))
    Cp, Ci = Cp.astype(np.int32), Ci.astype(np.int32)
    parent, cp = CO.schol(n, Cp, Ci)
    Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)          # the Cholesky factor in natural order: 4 810 levels for 4 884 columns
    L = cs.cs_pin(_host_cs(cs, n, n, Lp, Li, Lx))
    b = g["b"]
    for kind, fn in (("lsolve", CO.lsolve), ("ltsolve", CO.ltsolve)):
        x = b.tolist()
        alias = x
        assert getattr(cs, "cs_" + kind)(L, x) is True and alias is x
        assert np.asarray(x).tobytes() == fn(n, Lp, Li, Lx, b).tobytes()
        t, _ = _taken(cs, L._dev.plans[{"lsolve": cs.TRI_L, "ltsolve": cs.TRI_LT}[kind]], n, b)
        assert t == 1
    # upper kinds: U = L' stored with the diagonal last
    Tp, Ti, Tx = CO.transpose(n, n, Lp, Li, Lx)
    U = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    for kind, fn in (("usolve", CO.usolve), ("utsolve", CO.utsolve)):
        x = b.tolist()
        assert getattr(cs, "cs_" + kind)(U, x) is True
        assert np.asarray(x).tobytes() == fn(n, Tp, Ti, Tx, b).tobytes()
    # a device block on the same factor is not a list: the device solves it, same bits
    X = cs.dvec(np.stack([b, 2 * b], axis=1))
    assert cs.cs_lsolve(L, X) is True
    assert X.numpy().reshape(n, 2)[:, 0].tobytes() == CO.lsolve(n, Lp, Li, Lx, b).tobytes()


def test_a_factor_that_is_no_chain_stays_on_the_device(cs, host_chains):
    rng = np.random.default_rng(2)
    n, Tp, Ti, Tx = _block_tri(rng, 200, [30, 7], 0.3, True, True)
    T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    b = 1.0 + np.arange(n) / n
    x = b.tolist()
    assert cs.cs_lsolve(T, x) is True
    assert np.asarray(x).tobytes() == CO.lsolve(n, Tp, Ti, Tx, b).tobytes()
    t, buf = _taken(cs, T._dev.plans[cs.TRI_L], n, b)
    assert t == 0 and buf.tobytes() == b.tobytes()


@pytest.mark.parametrize("order", [0, 1])
def test_cs_cholsol_with_a_list_on_a_chain(cs, host_chains, order):
    """bcsstk16, natural order (a chain: the host takes the solve) and order 1 (a bushy tree: the device keeps it): the answer has
    the bits of the device's own exact solve either way."""
    import _csx
    g = golden("bcsstk16")
    Cm = cs.cs_pin(unpack(cs, g, "C"))
    b = g["b"]
    x = b.tolist()
    assert cs.cs_cholsol(order, Cm, x) is True
    cs.cs_option("tri.host_chains", 0)
    y = b.tolist()
    assert cs.cs_cholsol(order, Cm, y) is True
    cs.cs_option("tri.host_chains", 1)
    assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
    F = cs.cholsol_factor(Cm, order)
    z = b.tolist()
    assert F.solve(z) is True and np.asarray(z).tobytes() == np.asarray(y).tobytes()
    t = C.c_int(-1)
    buf = b.copy()
    _csx.check(_csx.lib().csx_cholsol_set_order(F.plan_handle, 1))
    _csx.check(_csx.lib().csx_cholsol_solve_list(F.plan_handle, _csx.pd(buf), t))
    assert t.value == (1 if order == 0 else 0)
