#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference module.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

It imports /root/reference/csparse.py, runs the hot-path functions the
reference can execute on the reference's own matrices
(/root/reference/matrix/*) and on small seeded synthetic cases, and stores
inputs + outputs as compressed numpy archives.  Nothing of the reference's
source is stored: only numbers.  The fixtures travel to the GPU box; the
reference does not.

Large outputs (bcsstk16 A*A') are stored as SHA-256 digests of their
little-endian int64 / float64 bytes plus the small arrays (p).
"""
import hashlib
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
import csparse as R  # noqa: E402  (the reference, unmodified)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def I(a):
    return np.asarray(a, dtype=np.int64)


def F(a):
    return np.asarray(a, dtype=np.float64)


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def pack(prefix, A, d):
    """Store a CSC cs object under keys prefix_{m,n,p,i,x}; i/x cut to nnz."""
    nnz = A.p[A.n]
    d[prefix + "_mn"] = I([A.m, A.n, A.nzmax, len(A.i), -1 if A.x is None else len(A.x)])
    d[prefix + "_p"] = I(A.p)
    d[prefix + "_i"] = I(A.i[:nnz])
    if A.x is not None:
        d[prefix + "_x"] = F(A.x[:nnz])


def mk(m, n, p, i, x):
    A = R.cs_spalloc(m, n, len(i), x is not None, False)
    A.p = list(p)
    A.i = list(i)
    A.x = None if x is None else list(x)
    return A


def clone(A):
    B = R.cs()
    B.nzmax, B.m, B.n, B.nz = A.nzmax, A.m, A.n, A.nz
    B.p, B.i = list(A.p), list(A.i)
    B.x = None if A.x is None else list(A.x)
    return B


def rhs(m):
    # csparse_test.py:123-127
    return [1.0 + float(i) / m for i in range(m)]


class _Diag(R.cs_ifkeep):
    def fkeep(self, i, j, aij, other):
        return i != j


class _Lower(R.cs_ifkeep):
    def fkeep(self, i, j, aij, other):
        return i >= j


class _Upper(R.cs_ifkeep):
    def fkeep(self, i, j, aij, other):
        return i <= j


def make_sym(A):
    # csparse_test.py:115-120: C = A + triu(A,1)'
    AT = R.cs_transpose(A, True)
    R.cs_fkeep(AT, _Diag(), None)
    return R.cs_add(A, AT, 1, 1)


def is_sym(A):
    # csparse_test.py:102-113 (1 upper, -1 lower, 0 otherwise)
    if A.m != A.n:
        return 0
    up = lo = True
    for j in range(A.n):
        for p in range(A.p[j], A.p[j + 1]):
            if A.i[p] > j:
                up = False
            if A.i[p] < j:
                lo = False
    return 1 if up else (-1 if lo else 0)


def get_problem(name, tol=1e-14):
    # csparse_test.py:174-205
    T = R.cs_load(os.path.join(REF, "matrix", name))
    A = R.cs_compress(T)
    R.cs_dupl(A)
    sym = is_sym(A)
    R.cs_dropzeros(A)
    R.cs_droptol(A, tol)
    C = make_sym(A) if sym else A
    return T, A, C, sym


def sorted_copy(A):
    return R.cs_transpose(R.cs_transpose(A, True), True)


def matrix_fixture(name, big=False):
    d = {}
    meta = {}
    T = R.cs_load(os.path.join(REF, "matrix", name))
    d["T_mn"] = I([T.m, T.n, T.nzmax, T.nz])
    d["T_i"], d["T_j"], d["T_x"] = I(T.i[:T.nz]), I(T.p[:T.nz]), F(T.x[:T.nz])
    A = R.cs_compress(T)
    pack("A", A, d)
    meta["normA"] = R.cs_norm(A)
    AT = R.cs_transpose(A, True)
    pack("AT", AT, d)
    meta["normAT"] = R.cs_norm(AT)
    ATp = R.cs_transpose(A, False)
    assert ATp.x is None and ATp.i[:ATp.p[ATp.n]] == AT.i[:AT.p[AT.n]]
    # gaxpy: y = y0 + A x with x = 1 + j/n, y0 = 0.5 - i/m
    x = [1.0 + float(j) / A.n for j in range(A.n)]
    y = [0.5 - float(i) / A.m for i in range(A.m)]
    d["gaxpy_x"], d["gaxpy_y0"] = F(x), F(y)
    assert R.cs_gaxpy(A, x, y)
    d["gaxpy_y"] = F(y)
    # A*A' (csparse_test.py:262)
    C = R.cs_multiply(A, AT)
    nnzC = C.p[C.n]
    meta["AAT"] = dict(m=C.m, n=C.n, nnz=nnzC, nzmax=C.nzmax, leni=len(C.i), norm=R.cs_norm(C),
                       sha_p=sha(I(C.p)), sha_i=sha(I(C.i[:nnzC])), sha_x=sha(F(C.x[:nnzC])))
    if not big:
        pack("AAT", C, d)
    else:
        d["AAT_p"] = I(C.p)
    # pattern-only product (values None on one side -> C.x None, csparse.py:1625)
    Cpat = R.cs_multiply(ATp, A) if not big else None
    if Cpat is not None:
        assert Cpat.x is None
        pack("ATA_pat", Cpat, d)
    # D = A*A' + norm*I  (csparse_test.py:263) -> known answers of Test1
    Eye = mk(C.m, C.m, range(C.m + 1), range(C.m), [1.0] * C.m)
    D = R.cs_add(C, Eye, 1, R.cs_norm(C))
    meta["D"] = dict(nnz=D.p[D.n], norm=R.cs_norm(D))
    # the Test2 problem (csparse_test.py:174): dupl, dropzeros, droptol, make_sym
    T2, A2, C2, sym = get_problem(name)
    meta["sym"] = sym
    meta["C"] = dict(m=C2.m, n=C2.n, nnz=C2.p[C2.n], norm=R.cs_norm(C2),
                     sha_p=sha(I(C2.p)), sha_i=sha(I(C2.i[:C2.p[C2.n]])), sha_x=sha(F(C2.x[:C2.p[C2.n]])))
    pack("C", C2, d)
    if C2.m == C2.n:
        n = C2.n
        b = rhs(n)
        d["b"] = F(b)
        # triangular parts with sorted columns: diagonal first (L) / last (U)
        S = sorted_copy(C2)
        Lo = clone(S)
        R.cs_fkeep(Lo, _Lower(), None)
        Up = clone(S)
        R.cs_fkeep(Up, _Upper(), None)
        diag_ok = all(Lo.p[j] < Lo.p[j + 1] and Lo.i[Lo.p[j]] == j and Lo.x[Lo.p[j]] != 0 and
                      Up.i[Up.p[j + 1] - 1] == j for j in range(n))
        meta["tri_ok"] = bool(diag_ok)
        if diag_ok:
            pack("Lo", Lo, d)
            pack("Up", Up, d)
            for nm, fn, M in (("lsolve", R.cs_lsolve, Lo), ("ltsolve", R.cs_ltsolve, Lo),
                              ("usolve", R.cs_usolve, Up), ("utsolve", R.cs_utsolve, Up)):
                v = list(b)
                assert fn(M, v)
                d["x_" + nm] = F(v)
        # LU solve with the unmodified reference, natural ordering only
        tol = 0.001 if sym else 1.0
        if n <= 1000:
            v = list(b)
            ok = R.cs_lusol(0, C2, v, tol)
            meta["lusol_ok"] = bool(ok)
            if ok:
                d["x_lusol"] = F(v)
                meta["lusol_norm_inf"] = max(abs(t) for t in v)
                # the reference's own (quirky) factors as tri-solve inputs
                S0 = R.cs_sqr(0, C2, False)
                N0 = R.cs_lu(C2, S0, tol)
                pack("refL", N0.L, d)
                pack("refU", N0.U, d)
                d["ref_pinv"] = I(N0.pinv)
                w = [0.0] * n
                R.cs_ipvec(N0.pinv, b, w, n)
                d["ref_lu_pb"] = F(w)
                R.cs_lsolve(N0.L, w)
                d["ref_lu_y"] = F(w)
                R.cs_usolve(N0.U, w)
                d["ref_lu_x"] = F(w)
        # symbolic pieces that do run in the reference (pin the host restatement)
        if sym:
            Cu = R.cs_symperm(C2, None, False)
            pack("symperm", Cu, d)
            parent = R.cs_etree(Cu, False)
            d["etree"] = I(parent)
            d["post"] = I(R.cs_post(parent, n))
            if n <= 100:
                w = [0] * n
                s = [0] * n
                tops, pats = [], []
                for k in range(n):
                    top = R.cs_ereach(Cu, k, parent, s, 0, w)
                    tops.append(top)
                    pats.extend(s[top:n])
                assert all(t >= 0 for t in w)
                d["ereach_top"] = I(tops)
                d["ereach_pat"] = I(pats)
    # cs_qrsol with the unmodified reference, natural ordering (csparse_test.py:454-462).  The
    # reference is right for square matrices; for m != n its row permutation collides
    # (csparse.py:2179-2182) and its answer differs from csparse_test.py's expectation.
    if max(C2.m, C2.n) <= 1000:
        bb = rhs(C2.m) + [0.0] * max(0, C2.n - C2.m)
        try:
            okq = R.cs_qrsol(0, C2, bb)
        except ZeroDivisionError:
            okq = None
        meta["qrsol_ok"] = okq
        if okq:
            d["x_qrsol"] = F(bb[:C2.n])
            meta["qrsol_norm_inf"] = max(abs(t) for t in bb[:C2.n])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return meta


def synthetic_fixture(seed):
    """Small random cases that hit the edge semantics: rectangular, unsorted
    columns, duplicate entries, empty columns, explicit zeros, pattern-only."""
    rng = random.Random(seed)
    d = {}
    cases = []
    for c, (m, n, nz, dup) in enumerate([(7, 5, 19, True), (5, 9, 23, True), (12, 12, 40, False),
                                         (1, 6, 4, False), (6, 1, 5, True), (30, 17, 160, True),
                                         (4, 4, 0, False)]):
        T = R.cs_spalloc(0, 0, 1, True, True)
        for _ in range(nz):
            i, j = rng.randrange(m), rng.randrange(n)
            v = rng.choice([0.0, 1.0, -2.5, rng.uniform(-3, 3), rng.uniform(-3, 3)])
            R.cs_entry(T, i, j, v)
            if dup and rng.random() < 0.2:
                R.cs_entry(T, i, j, rng.uniform(-1, 1))
        T.m, T.n = m, n
        A = R.cs_compress(T)
        pre = "c%d_" % c
        pack(pre + "A", A, d)
        AT = R.cs_transpose(A, True)
        pack(pre + "AT", AT, d)
        x = [rng.uniform(-2, 2) for _ in range(n)]
        y = [rng.uniform(-2, 2) for _ in range(m)]
        d[pre + "x"], d[pre + "y0"] = F(x), F(y)
        R.cs_gaxpy(A, x, y)
        d[pre + "y"] = F(y)
        pack(pre + "AAT", R.cs_multiply(A, AT), d)
        pack(pre + "ATA", R.cs_multiply(AT, A), d)
        Apat = clone(A)
        Apat.x = None
        P = R.cs_multiply(Apat, AT)
        assert P.x is None
        pack(pre + "AAT_pat", P, d)
        assert R.cs_multiply(A, A) is None or m == n
        cases.append([m, n, A.p[n]])
    d["cases"] = I(cases)
    # cs_cumsum / cs_scatter / cs_ipvec / cs_pvec known answers
    c = [rng.randrange(0, 9) for _ in range(37)]
    p = [0] * 38
    cc = list(c)
    d["cumsum_c"], d["cumsum_ret"] = I(c), I([R.cs_cumsum(p, cc, 37)])
    d["cumsum_p"], d["cumsum_c_out"] = I(p), I(cc)
    perm = list(range(23))
    rng.shuffle(perm)
    b = [rng.uniform(-1, 1) for _ in range(23)]
    xi, xp = [0.0] * 23, [0.0] * 23
    R.cs_ipvec(perm, b, xi, 23)
    R.cs_pvec(perm, b, xp, 23)
    d["perm"], d["perm_b"], d["ipvec"], d["pvec"] = I(perm), F(b), F(xi), F(xp)
    np.savez_compressed(os.path.join(OUT, "synthetic_%d.npz" % seed), **d)


def config2_fixture():
    """BASELINE config 2: cs_gaxpy on bcsstk16 as the reference's tests use it (csparse_test.py:525: the Test2
    problem matrix C = A + triu(A,1)', 4884 x 4884, 290 378 entries), run by the unmodified reference.
    C itself is already in bcsstk16.npz (keys C_*); this stores the vectors."""
    T, A, C, sym = get_problem("bcsstk16")
    assert sym and C.p[C.n] == 290378
    n = C.n
    x = [1.0 + float(j) / n for j in range(n)]
    y = [0.5 - float(i) / n for i in range(n)]
    d = {"x": F(x), "y0": F(y)}
    assert R.cs_gaxpy(C, x, y)
    d["y"] = F(y)
    d["sha_C"] = np.frombuffer(bytes.fromhex(sha(I(C.p)) + sha(I(C.i[:C.p[n]])) + sha(F(C.x[:C.p[n]]))), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "config2_bcsstk16.npz"), **d)
    return {"n": n, "nnz": C.p[n], "y_first": y[0], "y_last": y[-1]}


def updown_fixture():
    """cs_updown (csparse.py:2318-2365) runs unmodified; cs_chol does not, so the factor it is given
    comes from this repository's restated cs_chol (oracle/csparse_oracle.py) -- an INPUT; every expected output
    below is what the unmodified reference's cs_updown makes of it.  W is built as csparse_test.py:680-690 does
    (column n/2 of L, scaled by seeded random numbers)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import csparse_oracle as O
    d, meta = {}, {}
    rng = random.Random(20240613)
    for name in ("bcsstk01", "bcsstk16"):
        T, A, C, sym = get_problem(name)
        n = C.n
        Co = O.cs_spalloc(n, n, len(C.i), True, False)
        Co.p, Co.i, Co.x = list(C.p), list(C.i), list(C.x)
        S = O.cs_schol(0, Co)
        N = O.cs_chol(Co, S)
        L = mk(n, n, N.L.p, N.L.i, N.L.x)                 # a reference `cs` object holding that factor
        lnz = L.p[n]
        k = n // 2
        cnt = L.p[k + 1] - L.p[k]
        W = R.cs_spalloc(n, 1, n, True, False)
        W.p[0], W.p[1] = 0, cnt
        s0 = L.x[L.p[k]]
        for q in range(cnt):
            W.i[q] = L.i[L.p[k] + q]
            W.x[q] = s0 * rng.random()
        big = name == "bcsstk16"
        pre = name + "_"
        d[pre + "parent"] = I(S.parent)
        d[pre + "W_i"], d[pre + "W_x"] = I(W.i[:cnt]), F(W.x[:cnt])
        d[pre + "W2_x"] = F([v if q == 0 else 50.0 * v for q, v in enumerate(W.x[:cnt])])
        if not big:
            d[pre + "L_p"], d[pre + "L_i"], d[pre + "L_x"] = I(L.p), I(L.i[:lnz]), F(L.x[:lnz])
        ok_up = R.cs_updown(L, +1, W, S.parent)
        up = F(L.x[:lnz])
        ok_down = R.cs_updown(L, -1, W, S.parent)
        down = F(L.x[:lnz])
        # a downdate that is not positive definite: the reference stops part way and leaves L partly changed
        W2 = clone(W)
        W2.x = [v if q == 0 else 50.0 * v for q, v in enumerate(W.x)]   # first column passes, a later one fails
        ok_bad = R.cs_updown(L, -1, W2, S.parent)
        bad = F(L.x[:lnz])
        meta[name] = dict(n=n, lnz=lnz, k=k, ok_update=bool(ok_up), ok_downdate=bool(ok_down), ok_not_pd=bool(ok_bad),
                          sha_up=sha(up), sha_down=sha(down), sha_not_pd=sha(bad))
        if big:
            sel = list(range(0, lnz, max(1, lnz // 2000)))
            d[pre + "sample"] = I(sel)
            d[pre + "up_s"], d[pre + "down_s"], d[pre + "bad_s"] = up[sel], down[sel], bad[sel]
        else:
            d[pre + "up"], d[pre + "down"], d[pre + "bad"] = up, down, bad
    np.savez_compressed(os.path.join(OUT, "updown.npz"), **d)
    return meta


def sqr_fixture():
    """cs_sqr(0, A, True) -- the QR analysis: column elimination tree, column counts of R, cs_vcount -- runs
    unmodified on the square problem matrices C of the reference's tests; parent, cp, pinv, leftmost,
    m2, lnz, unz as the reference computes them.  (Rectangular matrices with rows left without a pivot hit the
    reference's D10 numbering and are not pinned here.)"""
    d, meta = {}, {}
    for name in ("t1", "bcsstk01", "west0067", "fs_183_1", "bcsstk16"):
        T, A, C, sym = get_problem(name)
        S = R.cs_sqr(0, C, True)
        assert S is not None and S.m2 == C.m
        pre = name + "_"
        d[pre + "parent"], d[pre + "cp"] = I(S.parent[:C.n]), I(S.cp[:C.n])
        d[pre + "pinv"], d[pre + "leftmost"] = I(S.pinv[:C.m + C.n]), I(S.leftmost[:C.m])
        meta[name] = dict(m=C.m, n=C.n, m2=int(S.m2), lnz=int(S.lnz), unz=int(S.unz))
    np.savez_compressed(os.path.join(OUT, "sqr_qr.npz"), **d)
    return meta


def counts_fixture():
    """cs_counts(A, parent, post, ata=True) -- the column counts of chol(A'A) -- runs unmodified (the ata=False
    branch does not, D6) on the problem matrices C of the reference's tests (csparse_test.py:174-205), rectangular ones included, with parent =
    cs_etree(A, True) and post = cs_post(parent, n) from the unmodified reference too."""
    d, meta = {}, {}
    for name in ("t1", "bcsstk01", "west0067", "ash219", "fs_183_1", "ibm32a", "ibm32b", "lp_afiro", "bcsstk16"):
        T, A0, A, sym = get_problem(name)       # the problem matrix C of the matrix's own fixture (<name>.npz, keys C_*)
        parent = R.cs_etree(A, True)
        post = R.cs_post(parent, A.n)
        cnt = R.cs_counts(A, parent, post, True)
        assert cnt is not None and len(cnt) >= A.n
        pre = name + "_"
        d[pre + "parent"], d[pre + "post"], d[pre + "count"] = I(parent[:A.n]), I(post[:A.n]), I(cnt[:A.n])
        meta[name] = dict(m=A.m, n=A.n, total=int(sum(cnt[:A.n])))
    np.savez_compressed(os.path.join(OUT, "counts_ata.npz"), **d)
    return meta


def batch_fixture():
    """Blocks of right-hand sides for the batched solvers (lusol_factor, qrsol_factor: factor once, many columns): the
    UNMODIFIED reference's cs_lusol(0, C, b, tol) and cs_qrsol(0, C, b) column by column on the square problem matrices of
    its tests.  Column r of B is rhs(n) * (1 + r / 2) + r (csparse_test.py:123-127 scaled and shifted)."""
    d, meta = {}, {}
    for name in ("t1", "bcsstk01", "west0067", "fs_183_1"):
        T, A, C, sym = get_problem(name)
        n, k = C.n, 4
        tol = 0.001 if sym else 1.0
        b0 = rhs(n)
        B = np.stack([F(b0) * (1.0 + 0.5 * r) + r for r in range(k)], axis=1)
        XL, XQ = np.empty((n, k)), np.empty((n, k))
        for r in range(k):
            v = B[:, r].tolist()
            assert R.cs_lusol(0, C, v, tol)
            XL[:, r] = v
            v = B[:, r].tolist()
            assert R.cs_qrsol(0, C, v)
            XQ[:, r] = v[:n]
        d[name + "_B"], d[name + "_x_lusol"], d[name + "_x_qrsol"] = B, XL, XQ
        meta[name] = dict(n=n, k=k, tol=tol)
    np.savez_compressed(os.path.join(OUT, "solve_batches.npz"), **d)
    return meta


def main():
    if sys.argv[1:] == ["batch"]:      # added to an existing fixture set without regenerating the others
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["solve_batches"] = batch_fixture()
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    if sys.argv[1:] == ["counts"]:     # added to an existing fixture set without regenerating the others
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["counts_ata"] = counts_fixture()
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    if sys.argv[1:] == ["sqr"]:
        print(json.dumps(sqr_fixture(), indent=1))
        return
    if sys.argv[1:] == ["updown"]:
        print(json.dumps(updown_fixture(), indent=1))
        return
    if sys.argv[1:] == ["config2"]:
        print("config2", config2_fixture())
        return
    if sys.argv[1:] == ["mbeacxc"]:
        # the tenth matrix of csparse_test.py (Test1 :381-394, Test2 :596-606): 492 x 490, 49 920 entries, rank deficient;
        # added to an existing fixture set without regenerating the others
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["mbeacxc"] = matrix_fixture("mbeacxc", big=True)
        print("mbeacxc", json.dumps(meta["mbeacxc"])[:400])
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    meta = {}
    for name in ("t1", "bcsstk01", "west0067", "ash219", "fs_183_1", "ibm32a", "ibm32b", "lp_afiro"):
        meta[name] = matrix_fixture(name)
        print(name, json.dumps(meta[name])[:200])
    meta["bcsstk16"] = matrix_fixture("bcsstk16", big=True)
    print("bcsstk16", json.dumps(meta["bcsstk16"])[:300])
    meta["mbeacxc"] = matrix_fixture("mbeacxc", big=True)
    print("mbeacxc", json.dumps(meta["mbeacxc"])[:300])
    synthetic_fixture(20240601)
    meta["config2_bcsstk16"] = config2_fixture()
    meta["updown"] = updown_fixture()
    meta["sqr_qr"] = sqr_fixture()
    meta["solve_batches"] = batch_fixture()
    meta["counts_ata"] = counts_fixture()
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
