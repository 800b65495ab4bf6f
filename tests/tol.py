"""Tolerances of the parity tests, written down once.

north_star: x[] within 1e-10 relative.  SURVEY 8d gives the measure for one vector against its reference:

    err = max_i |v_i - ref_i| / max(|ref_i|, eps * sum_i|terms|)

-- componentwise relative error, except that a component which cancelled far below the terms it was summed from is held
to the rounding of those terms (`componentwise`).  Where two DIFFERENT factorisations of the same matrix are compared (the
device's Cholesky against the reference's LU, QR against LU, the device's L against the oracle's L) the answers
legitimately differ by the conditioning of the problem: the bound is then `cross_bound(cond)` = max(1e-10, 8 cond_1(A) eps),
with cond_1 estimated from a sparse LU (`cond1`), instead of a round number."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = 2.0 ** -52
X_RTOL = 1e-10           # BASELINE.json north_star


def componentwise(v, ref, terms=None):
    """SURVEY 8d's measure; terms[i] = sum of the magnitudes of the terms ref[i] was formed from (None: |ref| alone)."""
    v, ref = np.asarray(v, float), np.asarray(ref, float)
    scale = np.abs(ref) if terms is None else np.maximum(np.abs(ref), EPS * np.asarray(terms, float))
    scale = np.where(scale > 0.0, scale, 1.0)
    return float(np.max(np.abs(v - ref) / scale)) if len(ref) else 0.0


def normwise(v, ref):
    ref = np.asarray(ref, float)
    return float(np.max(np.abs(np.asarray(v, float) - ref)) / np.max(np.abs(ref)))


def csc(n, p, i, x, m=None):
    p = np.asarray(p)
    return sp.csc_matrix((np.asarray(x, float)[:p[-1]], np.asarray(i)[:p[-1]], p), shape=(m or n, n))


def cond1(A):
    """1-norm condition estimate of a square sparse matrix (Hager / Higham on a sparse LU)."""
    A = sp.csc_matrix(A)
    lu = spla.splu(A)
    n = A.shape[0]
    inv = spla.LinearOperator((n, n), matvec=lu.solve, rmatvec=lambda b: lu.solve(b, trans="T"))
    return float(spla.onenormest(A) * spla.onenormest(inv))


def cross_bound(cond):
    return max(X_RTOL, 8.0 * cond * EPS)


def cond2_dense(A):
    """2-norm condition number of a small (rectangular) sparse matrix from a dense SVD, rank-deficient directions left out."""
    sv = np.linalg.svd(sp.csc_matrix(A).toarray(), compute_uv=False)
    sv = sv[sv > sv[0] * max(A.shape) * EPS]
    return float(sv[0] / sv[-1])


def lsq_bound(cond2):
    """two QR factorisations of one least-squares / minimum-norm problem: the solution moves with cond_2(A)^2 (Wedin)"""
    return max(X_RTOL, 8.0 * cond2 * cond2 * EPS)


def cholsolve_terms(n, Lp, Li, Lx, y, x):
    """sum|terms| of the LAST substitution of cs_lsolve + cs_ltsolve (csparse.py:1360-1364): x_i = (y_i - sum_j L_ji x_j) / L_ii."""
    L = csc(n, Lp, Li, Lx)
    d = L.diagonal()
    S = abs(L - sp.diags(d))
    return (np.abs(y) + S.T @ np.abs(x)) / np.abs(d)
