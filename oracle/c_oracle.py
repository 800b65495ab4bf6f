"""ctypes front-end of the plain-C oracle (oracle/oracle.c)  --  TEST
INFRASTRUCTURE, NOT PRODUCT.  numpy arrays in, numpy arrays out; int32 indices,
float64 values.  Builds oracle/_build/liboracle.so with gcc on first use."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)


def build():
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.co_cumsum.restype = C.c_longlong
    return _lib


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _pd(a):
    return None if a is None else a.ctypes.data_as(_dp)


def gaxpy(m, n, Ap, Ai, Ax, x, y):
    """y += A x (returns a new array; inputs untouched)."""
    Ap, Ai, Ax, x = _i(Ap), _i(Ai), _d(Ax), _d(x)
    y = _d(y).copy()
    st = lib().co_gaxpy(m, n, _pi(Ap), _pi(Ai), _pd(Ax), _pd(x), _pd(y))
    assert st == 0
    return y


def transpose(m, n, Ap, Ai, Ax):
    Ap, Ai = _i(Ap), _i(Ai)
    Ax = None if Ax is None else _d(Ax)
    nnz = int(Ap[n])
    Cp = np.zeros(m + 1, dtype=np.int32)
    Ci = np.zeros(nnz, dtype=np.int32)
    Cx = None if Ax is None else np.zeros(nnz, dtype=np.float64)
    st = lib().co_transpose(m, n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(Cp), _pi(Ci), _pd(Cx))
    assert st == 0
    return Cp, Ci, Cx


def multiply(m, k, n, Ap, Ai, Ax, Bp, Bi, Bx):
    Ap, Ai, Bp, Bi = _i(Ap), _i(Ai), _i(Bp), _i(Bi)
    Ax = None if Ax is None else _d(Ax)
    Bx = None if Bx is None else _d(Bx)
    Cp = np.zeros(n + 1, dtype=np.int32)
    ci, cx = _ip(), _dp()
    st = lib().co_multiply(m, k, n, _pi(Ap), _pi(Ai), _pd(Ax), _pi(Bp), _pi(Bi), _pd(Bx),
                           _pi(Cp), C.byref(ci), C.byref(cx))
    assert st == 0
    nnz = int(Cp[n])
    Ci = np.ctypeslib.as_array(ci, shape=(max(nnz, 1),))[:nnz].copy()
    Cx = None
    if Ax is not None and Bx is not None:
        Cx = np.ctypeslib.as_array(cx, shape=(max(nnz, 1),))[:nnz].copy()
        lib().co_free(cx)
    lib().co_free(ci)
    return Cp, Ci, Cx


def _tri(fn, n, Tp, Ti, Tx, x):
    Tp, Ti, Tx = _i(Tp), _i(Ti), _d(Tx)
    x = _d(x).copy()
    st = getattr(lib(), fn)(n, _pi(Tp), _pi(Ti), _pd(Tx), _pd(x))
    if st == 2:
        raise ZeroDivisionError("float division by zero")
    assert st == 0
    return x


def lsolve(n, Lp, Li, Lx, x):
    return _tri("co_lsolve", n, Lp, Li, Lx, x)


def ltsolve(n, Lp, Li, Lx, x):
    return _tri("co_ltsolve", n, Lp, Li, Lx, x)


def usolve(n, Up, Ui, Ux, x):
    return _tri("co_usolve", n, Up, Ui, Ux, x)


def utsolve(n, Up, Ui, Ux, x):
    return _tri("co_utsolve", n, Up, Ui, Ux, x)


def ipvec(p, b):
    b = _d(b)
    x = np.zeros_like(b)
    p = None if p is None else _i(p)
    assert lib().co_ipvec(_pi(p), _pd(b), _pd(x), len(b)) == 0
    return x


def pvec(p, b):
    b = _d(b)
    x = np.zeros_like(b)
    p = None if p is None else _i(p)
    assert lib().co_pvec(_pi(p), _pd(b), _pd(x), len(b)) == 0
    return x


def schol(n, Ap, Ai):
    """Natural-order symbolic Cholesky: (parent[n], cp[n+1])."""
    Ap, Ai = _i(Ap), _i(Ai)
    parent = np.zeros(max(n, 1), dtype=np.int32)
    cp = np.zeros(n + 1, dtype=np.int32)
    assert lib().co_schol(n, _pi(Ap), _pi(Ai), _pi(parent), _pi(cp)) == 0
    return parent[:n], cp


def chol(n, Cp, Ci, Cx, parent, cp):
    """Numeric up-looking Cholesky; returns (Lp, Li, Lx) or None if not SPD."""
    Cp, Ci, Cx, parent, cp = _i(Cp), _i(Ci), _d(Cx), _i(parent), _i(cp)
    lnz = int(cp[n])
    Lp = np.zeros(n + 1, dtype=np.int32)
    Li = np.zeros(max(lnz, 1), dtype=np.int32)
    Lx = np.zeros(max(lnz, 1), dtype=np.float64)
    st = lib().co_chol(n, _pi(Cp), _pi(Ci), _pd(Cx), _pi(parent), _pi(cp), _pi(Lp), _pi(Li), _pd(Lx))
    if st == 3:
        return None
    assert st == 0
    return Lp, Li[:lnz], Lx[:lnz]
