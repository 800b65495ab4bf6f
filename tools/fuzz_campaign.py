"""Run ON THE GPU BOX: a longer randomised campaign than tests/test_gpu_fuzz.py (many seeds, shapes drawn at random) for
the kernels rewritten in round 2: the radix sort behind cs_transpose, the one-pass cs_multiply kernels, csx_spsolve,
the list-level cs_gaxpy.  Everything is checked against the C / Python oracles.  Progress goes to gpurun_out/fuzz.log.
  python tools/fuzz_campaign.py [seconds] [first seed, default 0: a later run with another value draws other cases]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("csparse.py_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import csparse as cs
import c_oracle as CO
import csparse_oracle as O
import _csx
_csx.init()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "fuzz.log"), "w")


def say(*a):
    print(*a, file=log, flush=True)
    print(*a, flush=True)


def host(mod, m, n, p, i, x):
    A = mod.cs_spalloc(m, n, max(len(i), 1), True, False)
    A.p, A.i, A.x = np.asarray(p).tolist(), np.asarray(i).tolist(), np.asarray(x).tolist()
    return A


def ragged(rng, m, n, mean_len, maxlen=None):
    lens = rng.poisson(mean_len, size=n)
    lens[rng.random(n) < 0.1] = 0
    if maxlen is not None:
        lens = np.minimum(lens, maxlen)
    p = np.zeros(n + 1, np.int32)
    p[1:] = np.cumsum(lens)
    i = rng.integers(0, m, size=int(p[-1])).astype(np.int32)
    x = rng.uniform(-2, 2, size=int(p[-1]))
    return p, i, x


t_end = time.time() + budget
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
counts = {"transpose": 0, "multiply": 0, "spsolve": 0, "trisolve": 0, "cholesky": 0, "band_trisolve": 0, "band_cholesky": 0,
          "nd_cholesky": 0}


def random_band_lower(rng, n, band, keep):
    """strictly lower band with a share `keep` of its entries (the first sub-diagonal always), as scipy CSC."""
    import scipy.sparse as sp
    diags, offs = [], []
    for d in range(1, band + 1):
        v = rng.uniform(-1, 1, size=n - d)
        if d > 1:
            v = v * (rng.random(n - d) < keep)
        diags.append(v)
        offs.append(-d)
    M = sp.diags(diags, offs, shape=(n, n), format="csc")
    M.eliminate_zeros()
    return M


while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 5
    if seed % 7 == 0:  # banded chains too big for x to sit in LDS (window kernels), 1 .. 70 right-hand sides (wave-per-row
        import scipy.sparse as sp                       # kernels when the rows are long), all four kinds, bit for bit
        n = int(rng.choice([15400, 22000, 40000]))
        band = int(rng.choice([4, 60, 300]))
        keep = float(rng.choice([1.0, 0.5]))
        k = int(rng.choice([1, 3, 20, 70]))
        Lw = random_band_lower(rng, n, band, keep)
        L = (Lw + sp.diags(rng.uniform(2.0, 4.0, size=n) * (1 + band))).tocsc()
        L.sort_indices()
        U = L.T.tocsc()
        U.sort_indices()
        B = rng.uniform(-1, 1, size=(n, k))
        for fn, ofn, M in ((cs.cs_lsolve, CO.lsolve, L), (cs.cs_ltsolve, CO.ltsolve, L), (cs.cs_usolve, CO.usolve, U),
                           (cs.cs_utsolve, CO.utsolve, U)):
            Tp, Ti, Tx = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
            A = cs.cs_pin(host(cs, n, n, Tp, Ti, Tx))
            dB = cs.dvec(B if k > 1 else B[:, 0].copy())
            assert fn(A, dB) is True
            Xk = dB.numpy().reshape(n, k)
            for r in sorted({0, k // 2, k - 1}):
                assert Xk[:, r].tobytes() == ofn(n, Tp, Ti, Tx, B[:, r]).tobytes(), ("band trisolve", seed, fn.__name__, n, band, keep, k, r)
        counts["band_trisolve"] += 1
        continue
    if seed % 7 == 1:  # forests of cliques (csx_cholclique.hip): block-diagonal, random block sizes and densities, entries of
        sizes = rng.integers(1, int(rng.choice([9, 33, 65, 90])), size=int(rng.choice([1, 40, 600]))).tolist()   # the lower part shuffled,
        dens = float(rng.choice([1.0, 0.5, 0.15]))                 # now and then a duplicate or a block of more than 64 columns (the
        if rng.random() < 0.3:                                     # round 5: EQUAL blocks of 8 .. 64 columns now and then (the fused
            sizes = [int(rng.choice([8, 16, 32, 48, 64]))] * int(rng.choice([1, 40, 600]))   # factor -> plan path, the matrix-core kernels)
            dens = float(rng.choice([1.0, 1.0, 0.5]))
        n = int(sum(sizes))                                        # general path must take over): schol, chol bit for bit, cholsol
        cols_i, cols_x, a = [], [], 0
        for bsz in sizes:
            R = rng.uniform(-1.0, 1.0, (bsz, bsz))
            Bm = R @ R.T / bsz + bsz * np.eye(bsz)
            keep = np.triu(rng.uniform(size=(bsz, bsz)) < dens)
            keep = keep | keep.T
            if rng.random() < 0.8:
                keep[0, :] = keep[:, 0] = True                     # the block's factor is a clique; otherwise whatever comes
            keep[np.arange(bsz), np.arange(bsz)] = True
            for c in range(bsz):
                rws = np.nonzero(keep[:, c])[0]
                lo = rws[rws > c]
                rng.shuffle(lo)
                rws = np.concatenate([rws[rws <= c], lo])
                cols_i.append(rws + a); cols_x.append(Bm[rws, c] * keep[rws, c])
            a += bsz
        if rng.random() < 0.15 and n > 3:                          # a duplicate in some upper part: last one wins (csparse.py:594)
            c = int(rng.integers(1, n))
            if len(cols_i[c]) and cols_i[c][0] < c:
                cols_i[c] = np.concatenate([cols_i[c][:1], cols_i[c]]); cols_x[c] = np.concatenate([[0.125], cols_x[c]])
        Cp = np.zeros(n + 1, np.int32); Cp[1:] = np.cumsum([len(c) for c in cols_i])
        Ci, Cx = np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x)
        parent, cp = CO.schol(n, Cp, Ci)
        Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
        A = cs.cs_pin(host(cs, n, n, Cp, Ci, Cx))
        S = cs.cs_schol(0, A)
        assert S is not None and list(S.parent) == parent.tolist() and list(S.cp) == cp.tolist(), ("clique schol", seed, n)
        N = cs.cs_chol(A, S)
        assert N is not None, ("clique chol returned None", seed, n)
        lnz = int(Lp[-1])
        assert N.L.p == Lp.tolist() and N.L.i[:lnz] == Li.tolist(), ("clique chol pattern", seed, n)
        got = np.asarray(N.L.x[:lnz])
        path = _csx.C.c_int32(-1)
        _csx.check(_csx.lib().csx_chol_info(path, None))
        counts["chol_path_%d" % path.value] = counts.get("chol_path_%d" % path.value, 0) + 1   # 1 cliques, 2 small trees, 0 general
        if path.value == 1:
            assert got.tobytes() == Lx.tobytes(), ("clique chol values", seed, n)
        else:
            assert float(np.max(np.abs(got - Lx))) <= 1e-13 * max(1.0, float(np.max(np.abs(Lx)))), ("chol values", seed, n)
        k = int(rng.choice([1, 3, 70]))
        B = rng.uniform(-1, 1, size=(n, k))
        F = cs.cholsol_factor(A, exact=True)
        X = cs.dvec(B if k > 1 else B[:, 0].copy())
        assert F.solve(X) is True
        Xn = X.numpy().reshape(n, k)
        gLp, gLi, gLx = np.asarray(F.L.p, np.int32), np.asarray(F.L.i[:lnz], np.int32), np.asarray(F.L.x[:lnz])
        for r in sorted({0, k - 1}):
            z = CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B[:, r]))
            assert Xn[:, r].tobytes() == z.tobytes(), ("clique cholsol", seed, n, k, r)
        # round 5: the one-call factor (csx_cholsol_factor) in the order blocks are solved in -- matrix cores where the forest allows
        # (equal blocks: operands written by the block kernel; unequal blocks / small trees: dense by size class) -- and the same with
        # "chol.exact" = 0 (fused multiply-adds / the blocked factorisation on the matrix cores): x[] within 1e-10, L.x within 1e-13
        kk = max(k, int(rng.choice([9, 70])))
        B2 = rng.uniform(-1, 1, size=(n, kk))
        Z = np.stack([CO.ltsolve(n, gLp, gLi, gLx, CO.lsolve(n, gLp, gLi, gLx, B2[:, r])) for r in sorted({0, kk - 1})], axis=1)
        scale = np.maximum(np.abs(Z), 1e-3 * np.max(np.abs(Z)))
        for opt in (1, 0):
            with _csx.option("chol.exact", opt):
                F2 = cs.cholsol_factor(A)
                fpath = _csx.C.c_int32(-1)
                _csx.check(_csx.lib().csx_cholsol_factor_info(fpath, None, None, None))
                X2 = cs.dvec(B2)
                assert F2.solve(X2) is True
                X2n = X2.numpy().reshape(n, kk)[:, sorted({0, kk - 1})]
                err = float(np.max(np.abs(X2n - Z) / scale))
                assert err <= 1e-10, ("one-call cholsol", seed, n, kk, opt, fpath.value, err)
                lx2 = np.asarray(F2.L.x[:lnz])
                if opt == 1 and path.value == 1:
                    assert lx2.tobytes() == Lx.tobytes(), ("one-call L.x", seed, n)
                else:
                    assert float(np.max(np.abs(lx2 - Lx))) <= 1e-13 * max(1.0, float(np.max(np.abs(Lx)))), ("one-call L.x rounding", seed, n, opt)
                assert F2.L.p == Lp.tolist() and F2.L.i[:lnz] == Li.tolist(), ("one-call pattern", seed, n, opt)
                key = "factor_path_%d_%s" % (fpath.value, "mc" if F2.info()["matrix_cores"] else "sub")
                counts[key] = counts.get(key, 0) + 1
        counts["clique_cholesky"] = counts.get("clique_cholesky", 0) + 1
        continue
    if seed % 7 == 5:  # order 1 (nested dissection) on grids with random holes: supernodes as dense trapezoids, level hints from
        import scipy.sparse as sp                       # the tree, wave-per-row solves; L against the oracle on the permuted matrix
        gx, gy = int(rng.integers(20, 160)), int(rng.integers(20, 160))
        n = gx * gy
        Tx = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(gx, gx))
        Ty = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(gy, gy))
        G = (sp.kron(sp.identity(gy), Tx) + sp.kron(Ty, sp.identity(gx))).tocoo()
        keep = (G.row == G.col) | (rng.random(G.nnz) < float(rng.choice([1.0, 0.9])))
        Lw = sp.coo_matrix((G.data[keep & (G.row > G.col)], (G.row[keep & (G.row > G.col)], G.col[keep & (G.row > G.col)])), shape=(n, n))
        Sm = (Lw + Lw.T).tocsc()
        Sm = (Sm + sp.diags(np.asarray(abs(Sm).sum(axis=0)).ravel() + rng.uniform(0.01, 1.0, size=n))).tocsc()
        Sm.sort_indices()
        Cp, Ci, Cx = Sm.indptr.astype(np.int32), Sm.indices.astype(np.int32), Sm.data.astype(np.float64)
        A = cs.cs_pin(host(cs, n, n, Cp, Ci, Cx))
        S = cs.cs_schol(1, A)
        assert S is not None, ("nd schol", seed)
        Cperm = cs.cs_symperm(A, S.pinv, True)
        nz = Cperm.p[n]
        Pp, Pi, Px = np.asarray(Cperm.p, np.int32), np.asarray(Cperm.i[:nz], np.int32), np.asarray(Cperm.x[:nz], np.float64)
        parent, cp = CO.schol(n, Pp, Pi)
        assert list(S.parent) == parent.tolist() and list(S.cp) == cp.tolist(), ("nd schol tree", seed, gx, gy)
        Lp, Li, Lx = CO.chol(n, Pp, Pi, Px, parent, cp)
        N = cs.cs_chol(A, S)
        assert N is not None, ("nd chol None", seed, gx, gy)
        lnz = int(Lp[-1])
        assert N.L.p == Lp.tolist() and N.L.i[:lnz] == Li.tolist(), ("nd chol pattern", seed, gx, gy)
        got = np.asarray(N.L.x[:lnz])
        assert float(np.max(np.abs(got - Lx))) <= 1e-12 * float(np.max(np.abs(Lx))), ("nd chol values", seed, gx, gy)
        k = int(rng.choice([1, 3, 20, 70]))
        b = rng.uniform(-1, 1, size=(n, k))
        for exact in (True, False):
            F = cs.cholsol_factor(A, 1, exact=exact)
            X = cs.dvec(b if k > 1 else b[:, 0].copy())
            assert F.solve(X) is True
            Xn = X.numpy().reshape(n, k)
            for r in sorted({0, k - 1}):
                res = float(np.max(np.abs(Sm @ Xn[:, r] - b[:, r])))
                assert res <= 1e-9, ("nd cholsol residual", seed, gx, gy, k, exact, res)
        counts["nd_cholesky"] += 1
        continue
    if seed % 7 == 3:  # banded SPD matrices with a chain tree: the blocked dense-band cs_chol (band > 80) and the register
        import scipy.sparse as sp                       # window below, bit for bit against the plain-C oracle; cholsol both orders
        n = int(rng.choice([600, 2100, 5000]))
        band = int(rng.choice([30, 90, 200, 400]))
        band = min(band, n - 1)
        keep = float(rng.choice([1.0, 0.6]))
        Lw = random_band_lower(rng, n, band, keep)
        Sm = (Lw + Lw.T).tocsc()
        diag = np.asarray(abs(Sm).sum(axis=0)).ravel() + rng.uniform(1.0, 2.0, size=n)
        Sm = (Sm + sp.diags(diag)).tocsc()
        Sm.sort_indices()
        Cp, Ci, Cx = Sm.indptr.astype(np.int32), Sm.indices.astype(np.int32), Sm.data.astype(np.float64)
        parent, cp = CO.schol(n, Cp, Ci)
        Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
        A = cs.cs_pin(host(cs, n, n, Cp, Ci, Cx))
        S = cs.cs_schol(0, A)
        N = cs.cs_chol(A, S)
        assert N is not None, ("band chol returned None", seed, n, band)
        lnz = int(Lp[-1])
        assert N.L.p == Lp.tolist() and N.L.i[:lnz] == Li.tolist(), ("band chol pattern", seed, n, band)
        got = np.asarray(N.L.x[:lnz])
        chain = bool(np.all(parent[:-1] == np.arange(1, n)))
        full = lnz >= 0.5 * n * (band + 1)              # the dense-band kernels take mostly-full bands only
        if chain and n > 512 and (full or band <= 176):
            assert got.tobytes() == Lx.tobytes(), ("band chol bits", seed, n, band, keep)
        else:
            assert float(np.max(np.abs(got - Lx))) <= 1e-13 * max(1.0, float(np.max(np.abs(Lx)))), ("band chol values", seed, n, band)
        b = rng.uniform(-1, 1, size=(n, 2))
        z = [CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, b[:, r])) for r in range(2)]
        for exact in (True, False):
            F = cs.cholsol_factor(A, 0, exact=exact)
            X = cs.dvec(b)
            assert F.solve(X) is True
            Xn = X.numpy()
            for r in range(2):
                err = float(np.max(np.abs(Xn[:, r] - z[r]))) / max(1.0, float(np.max(np.abs(z[r]))))
                assert err <= (1e-13 if exact else 1e-10), ("band cholsol", seed, n, band, exact, err)
        counts["band_cholesky"] += 1
        continue
    if kind == 3:      # the four triangular solves, list and 3-column device block, bit for bit: random, banded (chain
        n = int(rng.choice([1, 3, 70, 500, 2500]))          # kernels), block-diagonal (component kernels)
        shape = int(rng.integers(0, 3))
        band = int(rng.choice([3, 40, 150]))
        nblk = max(1, n // int(rng.choice([4, 20, 67])))
        tri = {}
        for lower in (True, False):
            cols_i, cols_x = [], []
            for j in range(n):
                lo, hi = (j + 1, n) if lower else (0, j)
                if shape == 1:
                    lo, hi = (j + 1, min(n, j + band)) if lower else (max(0, j - band), j)
                elif shape == 2:
                    blk = n // nblk + 1
                    b0 = (j // blk) * blk
                    lo, hi = (j + 1, min(n, b0 + blk)) if lower else (b0, j)
                cand = np.arange(lo, hi)
                kk = len(cand) if shape == 1 else min(int(rng.poisson(3.0)), len(cand))
                off = rng.choice(cand, size=kk, replace=False) if kk else np.zeros(0, np.int64)
                if shape == 1 or rng.random() < 0.5:
                    off = np.sort(off)
                vals = rng.uniform(-1, 1, size=kk)
                d = float(rng.uniform(2.0, 4.0) * (1 + kk))
                cols_i.append(np.concatenate([[j], off]) if lower else np.concatenate([off, [j]]))
                cols_x.append(np.concatenate([[d], vals]) if lower else np.concatenate([vals, [d]]))
            Tp = np.zeros(n + 1, np.int32); Tp[1:] = np.cumsum([len(c) for c in cols_i])
            tri[lower] = (Tp, np.concatenate(cols_i).astype(np.int32), np.concatenate(cols_x))
        B = rng.uniform(-1, 1, size=(n, 3))
        pinned = {}
        for fn, ofn, lower in ((cs.cs_lsolve, CO.lsolve, True), (cs.cs_ltsolve, CO.ltsolve, True),
                               (cs.cs_usolve, CO.usolve, False), (cs.cs_utsolve, CO.utsolve, False)):
            Tp, Ti, Tx = tri[lower]
            M = cs.cs_pin(host(cs, n, n, Tp, Ti, Tx))
            pinned[fn] = M
            b = B[:, 0].tolist()
            assert fn(M, b) is True
            assert np.asarray(b).tobytes() == ofn(n, Tp, Ti, Tx, B[:, 0]).tobytes(), ("trisolve list", seed, fn.__name__, n, shape)
            dB = cs.dvec(B)
            assert fn(M, dB) is True
            Xk = dB.numpy().reshape(n, 3)
            for r in range(3):
                assert Xk[:, r].tobytes() == ofn(n, Tp, Ti, Tx, B[:, r]).tobytes(), ("trisolve block", seed, fn.__name__, n, shape, r)
            # round 5: the rounding-equal order of the same plan (csx_tri_set_order; matrix cores when the factor falls into
            # many small components -- shape 2 with small blocks): 20 right-hand sides against the oracle, then exact again
            B20 = rng.uniform(-1, 1, size=(n, 20))
            plan = M._dev.plans[{cs.cs_lsolve: cs.TRI_L, cs.cs_ltsolve: cs.TRI_LT, cs.cs_usolve: cs.TRI_U, cs.cs_utsolve: cs.TRI_UT}[fn]]
            _csx.check(_csx.lib().csx_tri_set_order(plan, 0))
            try:
                d20 = cs.dvec(B20)
                _csx.check(_csx.lib().csx_tri_solve(plan, d20.handle, 20))
                mc = _csx.C.c_int32(0)
                _csx.check(_csx.lib().csx_tri_order_info(plan, mc, None))
            finally:
                _csx.check(_csx.lib().csx_tri_set_order(plan, 1))
            X20 = d20.numpy().reshape(n, 20)
            for r in (0, 19):
                z = ofn(n, Tp, Ti, Tx, B20[:, r])
                err = float(np.max(np.abs(X20[:, r] - z))) / max(float(np.max(np.abs(z))), 1e-300)
                assert err <= 1e-11, ("trisolve rounding-equal", seed, fn.__name__, n, shape, mc.value, err)
            counts["tri_rounding_equal_mc%d" % mc.value] = counts.get("tri_rounding_equal_mc%d" % mc.value, 0) + 1
        # round 5, late: cs_lusol's solve phase as one call (csx_lusol_solve) on this L and U with random permutations, both orders:
        # the same bits as the four calls it replaces (permutations fused into the sweeps or not), the exact order the oracle's bits
        lib = _csx.lib()
        pl, pu = pinned[cs.cs_lsolve]._dev.plans[cs.TRI_L], pinned[cs.cs_usolve]._dev.plans[cs.TRI_U]
        k = int(rng.choice([9, 20, 64, 100, 128]))
        Bk = rng.uniform(-1, 1, size=(n, k))
        perms = []
        for want in (rng.random() < 0.8, rng.random() < 0.8):
            perms.append(rng.permutation(n).astype(np.int32) if want else None)
        hs = []
        for pp in perms:
            if pp is None:
                hs.append(_csx.H(0))
            else:
                h = _csx.new_handle()
                _csx.check(lib.csx_ivec_upload(_csx.pi(pp), n, h))
                hs.append(h)
        try:
            for exact in (1, 0):
                for plan in (pl, pu):
                    _csx.check(lib.csx_tri_set_order(plan, exact))
                b1, w1 = cs.dvec(Bk), cs.dvec(n, k)
                fused = _csx.C.c_int(-1)
                _csx.check(lib.csx_lusol_solve(pl, pu, hs[0], hs[1], b1.handle, w1.handle, k, _csx.C.byref(fused)))
                got = b1.numpy().reshape(n, k).copy()
                b2, x2 = cs.dvec(Bk), cs.dvec(n, k)
                _csx.check(lib.csx_permute_vec(hs[0], b2.handle, x2.handle, n, k, 1))
                _csx.check(lib.csx_tri_solve(pl, x2.handle, k))
                _csx.check(lib.csx_tri_solve(pu, x2.handle, k))
                _csx.check(lib.csx_permute_vec(hs[1], x2.handle, b2.handle, n, k, 1))
                assert b2.numpy().tobytes() == got.tobytes(), ("lusol_solve against the four calls", seed, n, shape, k, exact, fused.value)
                ident = np.arange(n)
                for r in (0, k - 1):
                    pb = np.empty(n)
                    pb[perms[0] if perms[0] is not None else ident] = Bk[:, r]
                    z = CO.usolve(n, *tri[False], CO.lsolve(n, *tri[True], pb))
                    want = np.empty(n)
                    want[perms[1] if perms[1] is not None else ident] = z
                    if exact:
                        assert got[:, r].tobytes() == want.tobytes(), ("lusol_solve exact", seed, n, shape, k, r)
                    else:
                        err = float(np.max(np.abs(got[:, r] - want))) / max(float(np.max(np.abs(want))), 1e-300)
                        assert err <= 1e-11, ("lusol_solve rounding-equal", seed, n, shape, k, r, fused.value, err)
                key = "lusol_solve_%s_fused%d" % ("exact" if exact else "rounding", fused.value)
                counts[key] = counts.get(key, 0) + 1
        finally:
            for plan in (pl, pu):
                lib.csx_tri_set_order(plan, 1)
            for h in hs:
                if h.value:
                    _csx.free(h)
        counts["trisolve"] += 1
        continue
    if kind == 4:      # cs_schol + cs_chol + cs_cholsol on random SPD matrices: banded, block-diagonal, scattered
        n = int(rng.choice([1, 6, 90, 400, 1500]))
        shape = int(rng.integers(0, 3))
        band = int(rng.choice([2, 25, 120]))
        blk = int(rng.choice([3, 16, 64]))
        rows, colsj, vals = [], [], []
        for j in range(n):
            if shape == 0:
                cand = np.arange(j + 1, min(n, j + band))
            elif shape == 1:
                cand = np.arange(j + 1, min(n, (j // blk + 1) * blk))
            else:
                cand = rng.choice(np.arange(j + 1, n), size=min(int(rng.poisson(1.5)), n - j - 1), replace=False) if j + 1 < n else np.zeros(0, np.int64)
            for i in cand:
                rows.append(int(i)); colsj.append(j); vals.append(float(rng.uniform(-1, 1)))
        import scipy.sparse as sp
        Lw = sp.coo_matrix((vals, (rows, colsj)), shape=(n, n))
        Sm = (Lw + Lw.T).tocsc()
        diag = np.asarray(abs(Sm).sum(axis=0)).ravel() + rng.uniform(1.0, 2.0, size=n)
        Sm = (Sm + sp.diags(diag)).tocsc()
        Sm.sort_indices()
        Cp, Ci, Cx = Sm.indptr.astype(np.int32), Sm.indices.astype(np.int32), Sm.data.astype(np.float64)
        parent, cp = CO.schol(n, Cp, Ci)
        Lp, Li, Lx = CO.chol(n, Cp, Ci, Cx, parent, cp)
        A = host(cs, n, n, Cp, Ci, Cx)
        S = cs.cs_schol(0, A)
        assert S is not None and list(S.parent) == parent.tolist() and list(S.cp) == cp.tolist(), ("schol", seed, n, shape)
        N = cs.cs_chol(A, S)
        assert N is not None, ("chol returned None", seed, n, shape)
        lnz = int(Lp[-1])
        assert N.L.p == Lp.tolist() and N.L.i[:lnz] == Li.tolist(), ("chol pattern", seed, n, shape)
        got = np.asarray(N.L.x[:lnz])
        assert float(np.max(np.abs(got - Lx))) <= 1e-13 * max(1.0, float(np.max(np.abs(Lx)))), ("chol values", seed, n, shape)
        b = rng.uniform(-1, 1, size=n)
        x = b.tolist()
        assert cs.cs_cholsol(0, A, x) is True
        y = CO.lsolve(n, Lp, Li, Lx, b)
        z = CO.ltsolve(n, Lp, Li, Lx, y)
        assert float(np.max(np.abs(np.asarray(x) - z))) <= 1e-10 * max(1.0, float(np.max(np.abs(z)))), ("cholsol", seed, n, shape)
        counts["cholesky"] += 1
        continue
    if kind == 0:      # transpose: shapes across one to four radix passes, all column-start paths
        m = int(rng.choice([1, 7, 255, 257, 5000, 70000, 300000, 17000000]))
        n = int(rng.choice([1, 3, 200, 4097, 30000]))
        mean = float(rng.choice([0.05, 0.7, 3.0, 40.0, 300.0]))
        if n * mean > 3e6:
            mean = 3e6 / n
        Ap, Ai, Ax = ragged(rng, m, n, mean)
        Rp, Ri, Rx = CO.transpose(m, n, Ap, Ai, Ax)
        lib = _csx.lib()
        hA, hT = _csx.new_handle(), _csx.new_handle()
        Ai_, Ax_ = (Ai, Ax) if len(Ai) else (np.zeros(1, np.int32), np.zeros(1))
        _csx.check(lib.csx_csc_upload(m, n, _csx.pi(Ap), _csx.pi(Ai_), _csx.pd(Ax_), hA))
        _csx.check(lib.csx_transpose(hA, 1, hT))
        nnz = int(Ap[-1])
        Tp, Ti, Tx = np.empty(m + 1, np.int32), np.empty(max(nnz, 1), np.int32), np.empty(max(nnz, 1), np.float64)
        _csx.check(lib.csx_csc_download(hT, _csx.pi(Tp), _csx.pi(Ti), _csx.pd(Tx)))
        _csx.free(hA); _csx.free(hT)
        assert np.array_equal(Tp, Rp) and np.array_equal(Ti[:nnz], Ri) and Tx[:nnz].tobytes() == Rx.tobytes(), ("transpose", seed, m, n, mean)
        counts["transpose"] += 1
    elif kind == 1:    # multiply: m above and below the dense-accumulator limit, narrow and wide columns
        m = int(rng.choice([50, 9000, 20000, 100000]))
        k = int(rng.choice([30, 800, 5000]))
        n = int(rng.choice([1, 60, 900]))
        Ap, Ai, Ax = ragged(rng, m, k, float(rng.choice([1.0, 6.0, 25.0, 60.0])), maxlen=int(rng.choice([8, 32, 200])))
        Bp, Bi, Bx = ragged(rng, k, n, float(rng.choice([0.5, 8.0, 40.0, 90.0])))
        Cp, Ci, Cx = CO.multiply(m, k, n, Ap, Ai, Ax, Bp, Bi, Bx)
        _, _, Sx = CO.multiply(m, k, n, Ap, Ai, np.abs(Ax), Bp, Bi, np.abs(Bx))
        C = cs.cs_multiply(host(cs, m, k, Ap, Ai, Ax), host(cs, k, n, Bp, Bi, Bx))
        nnz = int(Cp[-1])
        assert C.p == Cp.tolist() and C.i[:nnz] == Ci.tolist(), ("multiply pattern", seed, m, k, n)
        err = np.abs(np.asarray(C.x[:nnz]) - Cx) / np.maximum(Sx, 1e-300) if nnz else np.zeros(1)
        assert float(err.max()) < 1e-10, ("multiply values", seed, float(err.max()))
        counts["multiply"] += 1
    else:              # spsolve: random triangles, with and without pinv
        n = int(rng.choice([1, 5, 64, 400, 1500]))
        nb = int(rng.choice([1, 70, 600]))
        lower = bool(rng.integers(0, 2))
        cols_i, cols_x = [], []
        for j in range(n):
            lo, hi = (j + 1, n) if lower else (0, j)
            kk = min(int(rng.poisson(2.5)), hi - lo)
            off = rng.choice(np.arange(lo, hi), size=kk, replace=False) if kk else np.zeros(0, np.int64)
            vals = rng.uniform(-1, 1, size=kk)
            d = float(rng.uniform(2.0, 4.0))
            cols_i.append(np.concatenate([[j], off]) if lower else np.concatenate([off, [j]]))
            cols_x.append(np.concatenate([[d], vals]) if lower else np.concatenate([vals, [d]]))
        Gp = np.zeros(n + 1, np.int32); Gp[1:] = np.cumsum([len(c) for c in cols_i])
        Gi = np.concatenate(cols_i).astype(np.int32); Gx = np.concatenate(cols_x)
        Bp, Bi, Bx = ragged(rng, n, nb, 2.0)
        pinv = None
        if lower and rng.random() < 0.5:
            pinv = rng.permutation(n).astype(np.int32)
            pinv[rng.random(n) < 0.1] = -1
            pinv = pinv.tolist()
        oG, oB = host(O, n, n, Gp, Gi, Gx), host(O, n, nb, Bp, Bi, Bx)
        xi, x = [0] * (2 * n), [0.0] * n
        p, idx, val = [0], [], []
        for kcol in range(nb):
            top = O.cs_spsolve(oG, oB, kcol, xi, x, pinv, lower)
            idx += xi[top:n]; val += [x[j] for j in xi[top:n]]; p.append(len(idx))
        X = cs.spsolve_columns(host(cs, n, n, Gp, Gi, Gx), host(cs, n, nb, Bp, Bi, Bx), pinv, lower)
        assert X.p == p and X.i[:p[-1]] == idx, ("spsolve pattern", seed, n, nb)
        assert np.asarray(X.x[:p[-1]]).tobytes() == np.asarray(val, dtype=np.float64).tobytes(), ("spsolve values", seed)
        counts["spsolve"] += 1
    if seed % 10 == 0:
        say("seed", seed, counts)
say("done", counts)
