"""Parity of the device triangular solves and vector permutations against the
oracle and the golden vectors (needs an MI355X)."""
import numpy as np
import pytest

import c_oracle as CO
import csparse_oracle as O
import synth
from conftest import golden, unpack
from test_gpu_parity import ALL, _host_cs, cs  # noqa: F401  (cs is the module fixture)

pytestmark = pytest.mark.gpu


def _solvers(mod):
    return {"lsolve": mod.cs_lsolve, "ltsolve": mod.cs_ltsolve, "usolve": mod.cs_usolve, "utsolve": mod.cs_utsolve}


@pytest.mark.parametrize("name", ALL)
def test_trisolves_bit_exact_on_reference_matrices(cs, name):
    g = golden(name)
    if "x_lsolve" not in g:
        pytest.skip("no triangular fixture (rectangular or zero diagonal)")
    Lo, Up = unpack(cs, g, "Lo"), unpack(cs, g, "Up")
    for nm, fn in _solvers(cs).items():
        x = g["b"].tolist()
        alias = x
        assert fn(Lo if nm.startswith("l") else Up, x) is True
        assert alias is x and np.asarray(x).tobytes() == g["x_" + nm].tobytes(), nm


@pytest.mark.parametrize("name", ["t1", "bcsstk01", "west0067", "fs_183_1"])
def test_trisolves_on_the_reference_lu_factors(cs, name):
    """L and U as the reference's own cs_lu emits them: unsorted columns, duplicate rows,
    explicit zeros.  cs_lusol's solve sequence must come out bit-identical."""
    g = golden(name)
    L, U = unpack(cs, g, "refL"), unpack(cs, g, "refU")
    n = L.n
    b = g["b"].tolist()
    x = [0.0] * n
    assert cs.cs_ipvec(g["ref_pinv"].tolist(), b, x, n)
    assert x == g["ref_lu_pb"].tolist()
    assert cs.cs_lsolve(L, x) and np.asarray(x).tobytes() == g["ref_lu_y"].tobytes()
    assert cs.cs_usolve(U, x) and np.asarray(x).tobytes() == g["ref_lu_x"].tobytes()
    out = [0.0] * n
    assert cs.cs_ipvec(None, x, out, n) and np.asarray(out).tobytes() == g["x_lusol"].tobytes()


def test_trisolve_many_rhs_and_device_permutation(cs):
    g = golden("bcsstk16")
    n, k = 4884, 7
    Lo, Up = cs.cs_pin(unpack(cs, g, "Lo")), cs.cs_pin(unpack(cs, g, "Up"))
    B = synth.rhs(n, k, 3)
    oLo, oUp = unpack(O, g, "Lo"), unpack(O, g, "Up")
    for nm in ("lsolve", "ltsolve", "usolve", "utsolve"):
        X = cs.dvec(B)
        assert _solvers(cs)[nm](Lo if nm.startswith("l") else Up, X) is True
        got = X.numpy()
        for r in range(k):
            ref = B[:, r].tolist()
            _solvers(O)[nm](oLo if nm.startswith("l") else oUp, ref)
            assert got[:, r].tobytes() == np.asarray(ref).tobytes(), (nm, r)
    rng = np.random.default_rng(3)
    perm = rng.permutation(n).astype(np.int32)
    db, dx = cs.dvec(B), cs.dvec(n, k)
    assert cs.cs_ipvec(perm.tolist(), db, dx, n)
    ref = np.empty_like(B)
    ref[perm] = B
    assert dx.numpy().tobytes() == ref.tobytes()
    assert cs.cs_pvec(perm.tolist(), db, dx, n)
    assert dx.numpy().tobytes() == B[perm].tobytes()
    assert cs.cs_pvec(None, db, dx, n) and dx.numpy().tobytes() == B.tobytes()


def test_trisolve_zero_pivot_and_malformed(cs):
    L = cs.cs_spalloc(3, 3, 4, True, False)
    L.p, L.i, L.x = [0, 2, 3, 4], [0, 2, 1, 2], [2.0, 1.0, 0.0, 4.0]
    with pytest.raises(ZeroDivisionError):
        cs.cs_lsolve(L, [1.0, 1.0, 1.0])
    with pytest.raises(ZeroDivisionError):
        cs.cs_ltsolve(L, [1.0, 1.0, 1.0])
    # an entry above the diagonal in a non-first slot: the reference's push loop still runs
    # (it rewrites an already final unknown); the device must reproduce that, not "fix" it
    M = cs.cs_spalloc(4, 4, 7, True, False)
    M.p, M.i, M.x = [0, 2, 4, 6, 7], [0, 3, 1, 0, 2, 3, 3], [2.0, 0.5, 4.0, 0.25, 5.0, 1.5, 8.0]
    Mo = O.cs_spalloc(4, 4, 7, True, False)
    Mo.p, Mo.i, Mo.x = list(M.p), list(M.i), list(M.x)
    for nm in ("lsolve", "ltsolve"):
        x, ref = [1.0, 2.0, 3.0, 4.0], [1.0, 2.0, 3.0, 4.0]
        assert _solvers(cs)[nm](M, x) and _solvers(O)[nm](Mo, ref)
        assert x == ref, nm
    # a column with no entries has no diagonal
    E = cs.cs_spalloc(2, 2, 1, True, False)
    E.p, E.i, E.x = [0, 1, 1], [0], [1.0]
    with pytest.raises((ValueError, IndexError)):
        cs.cs_lsolve(E, [1.0, 1.0])


def _random_lower(n, per_col, seed):
    """Sparse lower-triangular CSC with the diagonal first in each column, rows unsorted."""
    rng = np.random.default_rng(seed)
    cols_i, cols_x, Ap = [], [], [0]
    for j in range(n):
        cnt = min(per_col, n - 1 - j)
        rows = (j + 1 + rng.choice(n - 1 - j, size=cnt, replace=False)) if cnt > 0 else np.empty(0, dtype=np.int64)
        cols_i.append(np.concatenate([[j], rows]))
        cols_x.append(np.concatenate([[2.0 + rng.random()], rng.uniform(-0.2, 0.2, cnt)]))
        Ap.append(Ap[-1] + cnt + 1)
    return (np.asarray(Ap, dtype=np.int32), np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x))


def test_trisolve_medium_random_against_c_oracle(cs):
    n = 6000
    Lp, Li, Lx = _random_lower(n, 6, 9)
    L = cs.cs_pin(_host_cs(cs, n, n, Lp, Li, Lx))
    b = synth.vec(n, 4, 0.5, 1.5)
    for nm, ref in (("lsolve", CO.lsolve(n, Lp, Li, Lx, b)), ("ltsolve", CO.ltsolve(n, Lp, Li, Lx, b))):
        x = cs.dvec(b)
        assert _solvers(cs)[nm](L, x)
        assert x.numpy().tobytes() == ref.tobytes(), nm
    # U: transposing the diagonal-first lower factor gives rows ascending with the diagonal last
    Up, Ui, Ux = CO.transpose(n, n, Lp, Li, Lx)
    U = cs.cs_pin(_host_cs(cs, n, n, Up, Ui, Ux))
    for nm, ref in (("usolve", CO.usolve(n, Up, Ui, Ux, b)), ("utsolve", CO.utsolve(n, Up, Ui, Ux, b))):
        x = cs.dvec(b)
        assert _solvers(cs)[nm](U, x)
        assert x.numpy().tobytes() == ref.tobytes(), nm
