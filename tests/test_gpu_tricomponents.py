"""Triangular solves on matrices whose dependency graph falls into many small connected components
(BASELINE config 3's W: 1 493 independent 67 x 67 blocks): one wave per component, X tile in LDS, no level
sets (csx_trisolve.hip: analyse_components / k_tri_local).  Bit-identical to the reference's loops for every
kind and every right-hand side, and equal to the level-scheduled path it replaces."""
import numpy as np
import pytest

import c_oracle as CO
import synth
from test_gpu_configs import _w_matrix
from test_gpu_parity import _host_cs, cs  # noqa: F401

pytestmark = pytest.mark.gpu


def _block_tri(rng, nb, sizes, density, lower, unit_first):
    """Block-diagonal triangular matrix with blocks of the given sizes (cycled), random pattern inside a block,
    diagonal first (lower) / last (upper) in every column, rows of a column otherwise in random order."""
    cols_i, cols_x, Ap = [], [], [0]
    base = 0
    for b in range(nb):
        m = sizes[b % len(sizes)]
        for c in range(m):
            cand = np.arange(c + 1, m) if lower else np.arange(0, c)
            pick = cand[rng.random(len(cand)) < density]
            rng.shuffle(pick)
            d = rng.uniform(1.0, 2.0) * (1 if rng.random() < 0.5 else -1)
            rows = ([c] + pick.tolist()) if lower else (pick.tolist() + [c])
            vals = ([d] + rng.uniform(-1, 1, len(pick)).tolist()) if lower else (rng.uniform(-1, 1, len(pick)).tolist() + [d])
            cols_i.append(np.asarray(rows, np.int32) + base)
            cols_x.append(np.asarray(vals))
            Ap.append(Ap[-1] + len(rows))
        base += m
    return base, np.asarray(Ap, np.int32), np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)


@pytest.mark.parametrize("nrhs", [1, 3, 8, 32, 33, 64, 130])
@pytest.mark.parametrize("kind", ["lsolve", "ltsolve", "usolve", "utsolve"])
def test_component_solves_bit_identical(cs, kind, nrhs):
    import _csx
    rng = np.random.default_rng(hash(kind) % 1000 + nrhs)
    lower = kind in ("lsolve", "ltsolve")
    n, Tp, Ti, Tx = _block_tri(rng, 300, [67, 1, 5, 130, 64, 2], 0.15, lower, True)
    T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
    B = synth.rhs(n, nrhs, 0) if nrhs > 1 else synth.rhs(n, 1, 0)[:, 0]
    ref_fn = getattr(CO, kind)
    Bm = B.reshape(n, -1)
    refs = np.stack([ref_fn(n, Tp, Ti, Tx, Bm[:, r]) for r in range(Bm.shape[1])], axis=1)
    X = cs.dvec(B)
    assert getattr(cs, "cs_" + kind)(T, X) is True
    got = X.numpy().reshape(n, -1)
    assert got.tobytes() == refs.tobytes()
    plan = T._dev.plans[{"lsolve": cs.TRI_L, "ltsolve": cs.TRI_LT, "usolve": cs.TRI_U, "utsolve": cs.TRI_UT}[kind]]
    comp = _csx.C.c_int32()
    _csx.check(_csx.lib().csx_tri_components(plan, comp))
    assert comp.value >= 300                       # it really took the component path (sparse blocks may split)
    # and the level-scheduled path (forced) gives the same bits
    with _csx.option("tri.components", 0):
        T2 = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
        X2 = cs.dvec(B)
        assert getattr(cs, "cs_" + kind)(T2, X2) is True
        assert X2.numpy().tobytes() == X.numpy().tobytes()


def test_one_big_component_keeps_level_scheduling(cs):
    import _csx
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "bcsstk16.npz"))
    from conftest import unpack
    L = cs.cs_pin(unpack(cs, g, "Lo"))
    x = g["b"].tolist()
    assert cs.cs_lsolve(L, x) is True
    assert np.asarray(x).tobytes() == g["x_lsolve"].tobytes()
    comp = _csx.C.c_int32()
    _csx.check(_csx.lib().csx_tri_components(L._dev.plans[cs.TRI_L], comp))
    assert comp.value == 0


def test_W_factors_take_the_component_path(cs):
    import _csx
    n, Ap, Ai, Ax = _w_matrix(1493)
    A = _host_cs(cs, n, n, Ap, Ai, Ax)
    N = cs.cs_lu(A, cs.cs_sqr(0, A, False), 1.0)
    L, U = cs.cs_pin(N.L), cs.cs_pin(N.U)
    b = 1.0 + np.arange(n) / n
    X = cs.dvec(b)
    assert cs.cs_lsolve(L, X) and cs.cs_usolve(U, X)
    Lp, Li, Lx = (np.asarray(v) for v in (L.p, L.i, L.x))
    Up, Ui, Ux = (np.asarray(v) for v in (U.p, U.i, U.x))
    ref = CO.usolve(n, Up.astype(np.int32), Ui.astype(np.int32), Ux, CO.lsolve(n, Lp.astype(np.int32), Li.astype(np.int32), Lx, b))
    assert X.numpy().tobytes() == ref.tobytes()
    for M, k in ((L, cs.TRI_L), (U, cs.TRI_U)):
        comp = _csx.C.c_int32()
        _csx.check(_csx.lib().csx_tri_components(M._dev.plans[k], comp))
        assert comp.value >= 1493                  # a block may itself fall apart


@pytest.mark.parametrize("kind", ["lsolve", "ltsolve", "usolve", "utsolve"])
def test_level_analysis_on_the_device_gives_the_same_plan(cs, kind):
    """Level sets and chain-walker tables computed on the device (the path big factors take) must give the same
    bits as the host analysis: forced here on a medium factor with the component path switched off."""
    import _csx
    rng = np.random.default_rng(7)
    lower = kind in ("lsolve", "ltsolve")
    n, Tp, Ti, Tx = _block_tri(rng, 40, [300, 17, 64], 0.05, lower, True)
    B = synth.rhs(n, 5, 0)
    ref = np.stack([getattr(CO, kind)(n, Tp, Ti, Tx, B[:, r]) for r in range(5)], axis=1)
    outs = {}
    for where in (1, 2):
        with _csx.option("tri.components", 0), _csx.option("tri.levels_where", where):
            T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
            X = cs.dvec(B)
            assert getattr(cs, "cs_" + kind)(T, X) is True
            outs[where] = X.numpy()
            plan = T._dev.plans[{"lsolve": cs.TRI_L, "ltsolve": cs.TRI_LT, "usolve": cs.TRI_U, "utsolve": cs.TRI_UT}[kind]]
            lv = _csx.C.c_int32()
            _csx.check(_csx.lib().csx_tri_info(plan, None, lv, None))
            outs[("levels", where)] = lv.value
    assert outs[1].tobytes() == ref.tobytes() and outs[2].tobytes() == ref.tobytes()
    assert outs[("levels", 1)] == outs[("levels", 2)] > 1


def test_device_level_analysis_gives_up_on_deep_chains(cs):
    """bcsstk16's factor has 4810 levels: deeper than the device pass is willing to go; it must hand over to the host
    pass and still produce the reference's bits."""
    import _csx
    from conftest import GOLDEN, unpack
    g = np.load(__import__("os").path.join(GOLDEN, "bcsstk16.npz"))
    with _csx.option("tri.levels_where", 2):
        L = cs.cs_pin(unpack(cs, g, "Lo"))
        x = g["b"].tolist()
        assert cs.cs_lsolve(L, x) is True
    assert np.asarray(x).tobytes() == g["x_lsolve"].tobytes()


def test_malformed_triangle_through_the_device_analysis(cs):
    """An entry above the diagonal in 'L': the reference still runs its loop; so must the device path."""
    import _csx
    n = 300
    rng = np.random.default_rng(3)
    _, Tp, Ti, Tx = _block_tri(rng, 1, [n], 0.05, True, True)
    Ti = Ti.copy()
    q = int(Tp[200]) + 1 if Tp[201] - Tp[200] > 1 else int(Tp[200])
    if Tp[201] - Tp[200] > 1:
        Ti[q] = 5                                 # row 5 in column 200: above the diagonal
    b = synth.rhs(n, 1, 0)[:, 0]
    ref = CO.lsolve(n, Tp, Ti, Tx, b)
    with _csx.option("tri.components", 0), _csx.option("tri.levels_where", 2):
        T = cs.cs_pin(_host_cs(cs, n, n, Tp, Ti, Tx))
        x = b.tolist()
        assert cs.cs_lsolve(T, x) is True
    assert np.asarray(x).tobytes() == ref.tobytes()
