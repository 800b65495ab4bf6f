"""csparse -- drop-in for rwl/CSparse.py's sparse-direct hot path on AMD MI355X.

Same names, same positional signatures, same `cs` objects (CSC: p, i, x; nz ==
-1) and the same error conventions as the reference module
(/root/reference/csparse.py, cited as csparse.py:N): bad arguments give False /
None / -1, a zero pivot in a triangular solve raises ZeroDivisionError, vectors
are updated in place, matrix results are new `cs` objects.

What runs where
  * The loops of the hot path -- cs_gaxpy, cs_transpose, cs_multiply,
    cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve, the numeric part of cs_chol,
    and the permute/solve sequence of cs_cholsol / cs_lusol -- run as hand-written
    HIP kernels (gfx950) behind the C ABI of libcsx.so (include/csx.h), reached
    through ctypes (_csx.py).  There is no CPU fallback: without the library or a
    GPU these functions raise.
  * List-based calls behave exactly like the reference: inputs are uploaded,
    the result is written back into the caller's list.  For these calls the
    kernels keep the reference's order of floating-point operations, so y / x
    come back bit-identical.
  * Large problems stay on the device: `cs_pin(A)` keeps a matrix resident (and
    caches its analysis), `dvec` is a device-resident vector / n-by-k block that
    every function here accepts in place of a list, and results of
    cs_transpose / cs_multiply on pinned inputs are device-backed `cs` objects
    whose p / i / x are only copied to the host when read.
  * Host-side glue that the reference also does in Python (triplet assembly,
    cs_cumsum, cs_scatter on lists, permutation of a single list) stays Python.
"""
import os
import weakref
from math import sqrt

import numpy as np

import _csx
from _hostglue import (CS_FLIP, CS_MARK, CS_MARKED, CS_UNFLIP, cs_dfs, cs_ereach, cs_reach,  # noqa: F401
                       cs_spsolve)
from _csx import (GAXPY_ATOMIC, GAXPY_AUTO, GAXPY_EXACT, GAXPY_TILED,  # noqa: F401
                  GAXPY_WAVE, TRI_L, TRI_LT, TRI_U, TRI_UT)

CS_VER = 1
CS_SUBVER = 0
CS_SUBSUB = 0
CS_DATE = "May 14, 2012"
CS_COPYRIGHT = "Copyright (C) Timothy A. Davis, 2006-2011"


# --------------------------------------------------------------- containers --

class _DevMatrix(object):
    """Owner of a libcsx CSC handle (+ cached triangular-solve plans)."""

    def __init__(self, handle):
        self.handle = handle
        self.plans = {}
        self.version = 0      # bumped when the values change in place (cs_updown): solvers built on it re-plan
        self._fin = weakref.finalize(self, _DevMatrix._release, handle, self.plans)

    @staticmethod
    def _release(handle, plans):
        for h in plans.values():
            _csx.free(h)
        _csx.free(handle)

    def info(self):
        m, n, nnz, hv = (_csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int())
        _csx.check(_csx.lib().csx_csc_info(self.handle, m, n, nnz, hv), "csx_csc_info")
        return m.value, n.value, nnz.value, bool(hv.value)


class cs(object):
    """Matrix in compressed-column or triplet form (csparse.py:37-54).

    p, i, x behave as plain attributes.  A matrix produced on the device keeps
    them there until they are first read."""

    def __init__(self):
        self.nzmax = 0
        self.m = 0
        self.n = 0
        self._p = []
        self._i = []
        self._x = []
        self.nz = 0
        self._dev = None       # _DevMatrix when a device copy exists
        self._lazy = False     # True: host lists not materialised yet
        self._pinned = False
        self._implicit = False  # device-resident because an operation produced it there, not because of cs_pin

    def _materialise(self):
        if self._lazy:
            m, n, nnz, hv = self._dev.info()
            p = np.empty(n + 1, dtype=np.int32)
            i = np.empty(max(nnz, 1), dtype=np.int32)
            x = np.empty(max(nnz, 1), dtype=np.float64) if hv else None
            _csx.check(_csx.lib().csx_csc_download(self._dev.handle, _csx.pi(p), _csx.pi(i), _csx.pd(x)),
                       "csx_csc_download")
            keep = self.nzmax
            self._p = p.tolist()
            self._i = i[:nnz].tolist() + [0] * (keep - nnz)
            self._x = None if x is None else x[:nnz].tolist() + [0.0] * (keep - nnz)
            self._lazy = False
            if self._implicit:
                # The caller now holds plain lists and may edit them in place (C.x[k] = v, the reference's
                # idiom), which nothing here can observe: the host lists become the only copy.  cs_pin(C)
                # keeps a result resident on purpose (and then cs_invalidate applies).
                self._dev = None
                self._pinned = False
                self._implicit = False

    def _touch(self):
        # host data assigned: a non-pinned device copy is stale
        if self._dev is not None and not self._lazy:
            self._dev = None
            self._pinned = False

    @property
    def p(self):
        self._materialise()
        return self._p

    @p.setter
    def p(self, v):
        self._materialise()
        self._p = v
        self._touch()

    @property
    def i(self):
        self._materialise()
        return self._i

    @i.setter
    def i(self, v):
        self._materialise()
        self._i = v
        self._touch()

    @property
    def x(self):
        self._materialise()
        return self._x

    @x.setter
    def x(self, v):
        self._materialise()
        self._x = v
        self._touch()


class css(object):
    """Symbolic Cholesky / LU / QR analysis (csparse.py:57-76)."""

    def __init__(self):
        self.pinv = []
        self.q = []
        self.parent = []
        self.cp = []
        self.leftmost = []
        self.m2 = 0
        self.lnz = 0
        self.unz = 0


class csn(object):
    """Numeric Cholesky / LU / QR factorisation (csparse.py:79-90)."""

    def __init__(self):
        self.L = None
        self.U = None
        self.pinv = []
        self.B = []


class dvec(object):
    """Device-resident float64 vector (k == 1) or n-by-k row-major block.

    dvec(data)            upload a sequence / numpy array (2-D arrays keep their shape)
    dvec(n, k=1)          zeros
    Accepted wherever the reference takes a list x / y / b."""

    def __init__(self, data, k=None, _handle=None):
        if _handle is not None:
            self.handle, self.n, self.k = _handle, int(data), int(k or 1)
        elif isinstance(data, (int, np.integer)):
            self.n, self.k = int(data), int(k or 1)
            h = _csx.new_handle()
            _csx.check(_csx.lib().csx_vec_alloc(self.n * self.k, h), "csx_vec_alloc")
            self.handle = h
        else:
            a = _csx.f64(data)
            self.n = a.shape[0]
            self.k = 1 if a.ndim == 1 else int(np.prod(a.shape[1:]))
            h = _csx.new_handle()
            _csx.check(_csx.lib().csx_vec_upload(_csx.pd(a), a.size, h), "csx_vec_upload")
            self.handle = h
        self._fin = weakref.finalize(self, _csx.free, self.handle)

    def __len__(self):
        return self.n

    def numpy(self):
        out = np.empty(self.n * self.k, dtype=np.float64)
        _csx.check(_csx.lib().csx_vec_download(self.handle, _csx.pd(out), out.size), "csx_vec_download")
        return out if self.k == 1 else out.reshape(self.n, self.k)

    def tolist(self):
        return self.numpy().tolist()

    def assign(self, data):
        a = _csx.f64(data)
        _csx.check(_csx.lib().csx_vec_write(self.handle, _csx.pd(a), a.size), "csx_vec_write")

    def fill(self, value):
        _csx.check(_csx.lib().csx_vec_fill(self.handle, float(value)), "csx_vec_fill")

    def copy(self):
        out = dvec(self.n, self.k)
        _csx.check(_csx.lib().csx_vec_copy(self.handle, out.handle), "csx_vec_copy")
        return out

    def device_ptr(self):
        p = _csx.C.c_void_p()
        ln = _csx.C.c_int64()
        _csx.check(_csx.lib().csx_vec_ptr(self.handle, p, ln), "csx_vec_ptr")
        return p.value


# ------------------------------------------------------- predicates, alloc --

def CS_CSC(A):
    """True if A is compressed-column (csparse.py:113-119)."""
    return A is not None and A.nz == -1


def CS_TRIPLET(A):
    """True if A is a triplet matrix (csparse.py:122-128)."""
    return A is not None and A.nz >= 0


def ialloc(n):
    return [0] * n


def xalloc(n):
    return [0.0] * n


def cs_spalloc(m, n, nzmax, values, triplet):
    """Allocate a CSC or triplet matrix (csparse.py:2388-2406)."""
    A = cs()
    A.m = m
    A.n = n
    A.nzmax = nzmax = max(nzmax, 1)
    A.nz = 0 if triplet else -1
    A.p = ialloc(nzmax) if triplet else ialloc(n + 1)
    A.i = ialloc(nzmax)
    A.x = xalloc(nzmax) if values else None
    return A


def cs_sprealloc(A, nzmax):
    """Resize i / x (and p of a triplet); nzmax <= 0 trims (csparse.py:2414-2440)."""
    if A is None:
        return False
    if nzmax <= 0:
        nzmax = A.p[A.n] if CS_CSC(A) else A.nz
    A.i = (list(A.i) + [0] * nzmax)[:nzmax]
    if CS_TRIPLET(A):
        A.p = (list(A.p) + [0] * nzmax)[:nzmax]
    if A.x is not None:
        A.x = (list(A.x) + [0.0] * nzmax)[:nzmax]
    A.nzmax = nzmax
    return True


# ------------------------------------------- host glue (Python in the reference too) --

def cs_entry(T, i, j, x):
    """Append one triplet (csparse.py:1068-1091)."""
    if not CS_TRIPLET(T) or i < 0 or j < 0:
        return False
    if T.nz >= T.nzmax:
        cs_sprealloc(T, 2 * T.nzmax)
    if T.x is not None:
        T.x[T.nz] = x
    T.i[T.nz] = i
    T.p[T.nz] = j
    T.nz += 1
    T.m = max(T.m, i + 1)
    T.n = max(T.n, j + 1)
    return True


def cs_load(filename, base=0):
    """Read 'i j aij' lines into a triplet matrix (csparse.py:1307-1327)."""
    T = cs_spalloc(0, 0, 1, True, True)
    with open(filename, "rb") as fd:
        for line in fd:
            tok = line.split()
            if len(tok) != 3:
                return None
            if not cs_entry(T, int(tok[0]) - base, int(tok[1]) - base, float(tok[2])):
                return None
    return T


def cs_compress(T):
    """Triplet -> CSC, a stable sort by column (csparse.py:647-673), done with numpy."""
    if not CS_TRIPLET(T):
        return None
    nz = T.nz
    if T._pinned:   # cs_pin(T): sort on the device and keep the result there
        rows, cols = _csx.i32(T.i[:nz]), _csx.i32(T.p[:nz])
        vals = None if T.x is None else _csx.f64(T.x[:nz])
        h = _csx.new_handle()
        st = _csx.lib().csx_compress(T.m, T.n, nz, _csx.pi(rows), _csx.pi(cols), _csx.pd(vals), h)
        if st == _csx.EINVAL:
            raise IndexError("list index out of range")
        _csx.check(st, "csx_compress")
        return _from_device(h, lambda nnz: max(nnz, 1))
    C = cs_spalloc(T.m, T.n, nz, T.x is not None, False)
    cols = np.asarray(T.p[:nz], dtype=np.int64)
    order = np.argsort(cols, kind="stable")
    counts = np.bincount(cols, minlength=T.n) if nz else np.zeros(T.n, dtype=np.int64)
    C.p = [0] + np.cumsum(counts).tolist()
    pad = C.nzmax - nz
    C.i = np.asarray(T.i[:nz], dtype=np.int64)[order].tolist() + [0] * pad
    if T.x is not None:
        C.x = np.asarray(T.x[:nz], dtype=np.float64)[order].tolist() + [0.0] * pad
    return C


def cs_cumsum(p, c, n):
    """p[0..n] = exclusive prefix sums of c; c[0..n-1] = p[0..n-1] (csparse.py:767-784)."""
    if p is None or c is None:
        return -1
    total = 0
    for k in range(n):
        p[k] = total
        total += c[k]
        c[k] = p[k]
    p[n] = total
    return total


def cs_scatter(A, j, beta, w, x, mark, C, nz):
    """x += beta * A(:,j) on host lists, new rows appended to C.i (csparse.py:1961-1989).
    The device SpGEMM (cs_multiply) carries its own accumulator; this is the
    list-level primitive for callers that use it directly."""
    if not CS_CSC(A) or w is None or not CS_CSC(C):
        return -1
    Ai, Ax, Ci = A.i, A.x, C.i
    for p in range(A.p[j], A.p[j + 1]):
        r = Ai[p]
        if w[r] < mark:
            w[r] = mark
            Ci[nz] = r
            nz += 1
            if x is not None:
                x[r] = beta * Ax[p]
        elif x is not None:
            x[r] += beta * Ax[p]
    return nz


def cs_norm(A):
    """1-norm = largest column sum of |a| (csparse.py:1647-1663)."""
    if not CS_CSC(A):
        return -1
    if A._dev is not None:   # device-resident: column sums in storage order, the reference's bits
        if not A._dev.info()[3]:
            return -1
        out = _csx.C.c_double(0.0)
        _csx.check(_csx.lib().csx_norm1(A._dev.handle, out), "csx_norm1")
        return out.value
    if A.x is None:
        return -1
    best = 0
    for j in range(A.n):
        s = 0
        for p in range(A.p[j], A.p[j + 1]):
            s += abs(A.x[p])
        best = max(best, s)
    return best


def cs_pinv(p, n):
    """Inverse permutation (csparse.py:1696-1708)."""
    if p is None:
        return None
    inv = [0] * n
    for k in range(n):
        inv[p[k]] = k
    return inv


# ------------------------------------------------------------ device plumbing --

def _upload(A):
    """Host lists of a CSC `cs` -> libcsx handle.  IndexError for indices the
    reference would have tripped over."""
    n = A.n
    p = _csx.i32(A.p[:n + 1])
    nnz = int(p[n]) if n >= 0 and len(p) == n + 1 else -1
    if nnz < 0 or len(A.i) < nnz or (A.x is not None and len(A.x) < nnz):
        raise IndexError("list index out of range")
    i = _csx.i32(A.i[:nnz])
    x = None if A.x is None else _csx.f64(A.x[:nnz])
    h = _csx.new_handle()
    st = _csx.lib().csx_csc_upload(A.m, n, _csx.pi(p), _csx.pi(i), _csx.pd(x), h)
    if st == _csx.EINVAL:
        raise IndexError("list index out of range")
    _csx.check(st, "csx_csc_upload")
    return h


class _Resident(object):
    """Context manager: a device handle for A, temporary unless A is pinned."""

    def __init__(self, A):
        self.A = A
        self.temp = None

    def __enter__(self):
        A = self.A
        if A._dev is not None:
            return A._dev
        dev = _DevMatrix(_upload(A))
        if A._pinned:
            A._dev = dev
        else:
            self.temp = dev
        return dev

    def __exit__(self, *exc):
        if self.temp is not None:
            self.temp._fin()
        return False


def cs_pin(A):
    """Keep A resident on the device (and cache its analyses) until cs_unpin /
    its host lists are reassigned.  In-place edits of A.p / A.i / A.x after
    pinning are not seen: call cs_invalidate(A)."""
    if CS_TRIPLET(A):   # nothing to upload yet: cs_compress will sort it on the device and keep it there
        A._pinned = True
        return A
    if not CS_CSC(A):
        return None
    if A._dev is None:
        A._dev = _DevMatrix(_upload(A))
    A._pinned = True
    A._implicit = False
    return A


def cs_invalidate(A):
    """Tell the library that A.p / A.i / A.x were edited in place after cs_pin: the device copy and every plan
    cached on it (SpMV plans, triangular-solve plans) are dropped and rebuilt from the lists on the next use."""
    if A is not None and not A._lazy:
        A._dev = None
    return A


def cs_unpin(A):
    if A is not None:
        A._pinned = False
        if not A._lazy:
            A._dev = None
    return A


def _from_device(handle, nzmax_rule):
    """Wrap a device result as a lazily materialised `cs`."""
    dev = _DevMatrix(handle)
    m, n, nnz, hv = dev.info()
    C = cs()
    C.m, C.n, C.nz = m, n, -1
    C.nzmax = nzmax_rule(nnz)
    C._dev = dev
    C._lazy = True
    C._pinned = True
    C._implicit = True
    return C


def _vec_in(v, need, what):
    """list / numpy / dvec -> (dvec, writeback)"""
    if isinstance(v, dvec):
        if v.n * v.k < need:
            raise IndexError("list index out of range")
        return v, None
    if len(v) < need:
        raise IndexError("list index out of range")
    return dvec(np.asarray(v, dtype=np.float64)), v


def _write_back(host, d, count):
    if host is not None:
        out = d.numpy().reshape(-1)[:count]
        if isinstance(host, np.ndarray):
            host.reshape(-1)[:count] = out
        else:
            host[:count] = out.tolist()


# -------------------------------------------------------------- hot path ----

def cs_gaxpy(A, x, y, mode=None):
    """y = A*x + y (csparse.py:1199-1213).  True on success, False on bad input.

    Lists: y is updated in place with the reference's exact summation order.
    dvec x / y: stays on the device; `mode` picks the kernel (default: exact for
    list calls, the matrix's best plan for device calls)."""
    if not CS_CSC(A) or x is None or y is None:
        return False
    if not _meta(A)[1]:     # pattern only (asked of the device for a device-backed A: no download)
        raise TypeError("'NoneType' object is not subscriptable")
    dx, _ = _vec_in(x, A.n, "x")
    dy, yhost = _vec_in(y, A.m, "y")
    if mode is None:
        mode = GAXPY_EXACT if (yhost is not None or not isinstance(x, dvec)) else GAXPY_AUTO
    with _Resident(A) as dA:
        _csx.check(_csx.lib().csx_gaxpy(dA.handle, dx.handle, dy.handle, mode), "csx_gaxpy")
    _write_back(yhost, dy, A.m)
    return True


def cs_gaxpy_prepare(A, mode=GAXPY_AUTO):
    """Build the SpMV plan of a pinned matrix ahead of time (outside timed regions)."""
    cs_pin(A)
    _csx.check(_csx.lib().csx_gaxpy_prepare(A._dev.handle, mode), "csx_gaxpy_prepare")
    return True


def cs_transpose(A, values):
    """C = A' (csparse.py:2292-2315); None if A is not CSC."""
    if not CS_CSC(A):
        return None
    with _Resident(A) as dA:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_transpose(dA.handle, 1 if values else 0, h), "csx_transpose")
    C = _from_device(h, lambda nnz: max(nnz, 1))
    if not A._pinned:
        C._materialise()
        C._dev = None
        C._pinned = False
    return C


def cs_multiply(A, B):
    """C = A*B (csparse.py:1608-1642): columns in first-touch order, trimmed to nnz."""
    if not CS_CSC(A) or not CS_CSC(B):
        return None
    if A.n != B.m:
        return None
    with _Resident(A) as dA, _Resident(B) as dB:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_multiply(dA.handle, dB.handle, h), "csx_multiply")
    C = _from_device(h, lambda nnz: nnz)
    if not (A._pinned and B._pinned):
        C._materialise()
        C._i = C._i[:C.nzmax]
        if C._x is not None:
            C._x = C._x[:C.nzmax]
        C._dev = None
        C._pinned = False
    return C



def _meta(A):
    """(number of entries, has values) of a CSC matrix without pulling a device-resident one to the host."""
    if A._lazy:
        _, _, nnz, hv = A._dev.info()
        return nnz, hv
    return A._p[A.n], A._x is not None


def _result(h, nzmax_rule, keep_on_device):
    """Device result -> cs: lazy and pinned when the inputs were pinned, host lists otherwise."""
    C = _from_device(h, nzmax_rule)
    if not keep_on_device:
        C._materialise()
        C._i = (C._i + [0] * C.nzmax)[:C.nzmax]
        if C._x is not None:
            C._x = (C._x + [0.0] * C.nzmax)[:C.nzmax]
        C._dev = None
        C._pinned = False
    return C


def _replace_in_place(A, C):
    """Give A the contents of C (the reference's in-place functions mutate their argument)."""
    A.nzmax = C.nzmax
    if C._lazy:
        A._p, A._i, A._x = [], [], []
        A._dev, A._lazy, A._pinned, A._implicit = C._dev, True, True, C._implicit
    else:
        A._dev, A._lazy, A._pinned = None, False, False
        A._p, A._i, A._x = C._p, C._i, C._x


def cs_add(A, B, alpha, beta):
    """C = alpha*A + beta*B (csparse.py:163-192): column j in first-touch order over A(:,j) then
    B(:,j); not trimmed (nzmax = nnz(A) + nnz(B)).  None if not CSC or the shapes differ."""
    if not CS_CSC(A) or not CS_CSC(B):
        return None
    if A.m != B.m or A.n != B.n:
        return None
    room = _meta(A)[0] + _meta(B)[0]
    with _Resident(A) as dA, _Resident(B) as dB:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_add(dA.handle, dB.handle, float(alpha), float(beta), h), "csx_add")
    return _result(h, lambda nnz: max(room, 1), A._pinned and B._pinned)


def cs_dupl(A):
    """Sum duplicate entries into their first occurrence, in place (csparse.py:1035-1065)."""
    if not CS_CSC(A):
        return False
    nnz, hv = _meta(A)
    if not hv and nnz > 0:
        raise TypeError("'NoneType' object is not subscriptable")   # as the reference on a pattern-only matrix
    pinned = A._pinned
    with _Resident(A) as dA:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_dupl(dA.handle, h), "csx_dupl")
    _replace_in_place(A, _result(h, lambda nnz: nnz, pinned))
    return True


def cs_fkeep(A, fkeep, other):
    """Keep the entries for which fkeep(i, j, aij, other) is true, in place; returns the new number of
    entries, -1 on bad input (csparse.py:1172-1196).  The predicate is a Python callable, so this generic
    form runs on the host; cs_dropzeros / cs_droptol are the device versions of its two uses."""
    if not CS_CSC(A) or fkeep is None:
        return -1
    Ap, Ai, Ax, n = A.p, A.i, A.x, A.n
    nz = 0
    for j in range(n):
        p = Ap[j]
        Ap[j] = nz
        while p < Ap[j + 1]:
            if fkeep(Ai[p], j, Ax[p] if Ax is not None else 1.0, other):
                if Ax is not None:
                    Ax[nz] = Ax[p]
                Ai[nz] = Ai[p]
                nz += 1
            p += 1
    Ap[n] = nz
    A.p = Ap
    A.i = (Ai + [0] * nz)[:nz] if isinstance(Ai, list) else list(Ai[:nz])
    if Ax is not None:
        A.x = (Ax + [0.0] * nz)[:nz] if isinstance(Ax, list) else list(Ax[:nz])
    A.nzmax = nz
    return nz


def _drop(A, mode, tol):
    if not CS_CSC(A):
        return -1
    if not _meta(A)[1]:   # the reference passes aij = 1 for a pattern-only matrix
        keep_all = True if mode == 0 else 1.0 > tol
        return cs_fkeep(A, lambda i, j, a, o: keep_all, None)
    pinned = A._pinned
    with _Resident(A) as dA:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_drop(dA.handle, mode, float(tol), h), "csx_drop")
    C = _result(h, lambda nnz: nnz, pinned)
    _replace_in_place(A, C)
    return A._dev.info()[2] if A._lazy else A._p[A.n]


def cs_dropzeros(A):
    """Remove explicit zeros, in place; returns the new nnz (csparse.py:1019-1031)."""
    return _drop(A, 0, 0.0)


def cs_droptol(A, tol):
    """Remove entries with |a| <= tol, in place; returns the new nnz (csparse.py:1002-1014)."""
    return _drop(A, 1, tol)


def cs_permute(A, pinv, q, values):
    """C = P A Q, pinv the inverse row permutation, q the column permutation (csparse.py:1666-1693)."""
    if not CS_CSC(A):
        return None
    if pinv is not None and len(pinv) < A.m or q is not None and len(q) < A.n:
        raise IndexError("list index out of range")
    pv = None if pinv is None else _csx.i32(pinv)
    qv = None if q is None else _csx.i32(q)
    with _Resident(A) as dA:
        h = _csx.new_handle()
        st = _csx.lib().csx_permute(dA.handle, _csx.pi(pv), _csx.pi(qv), 1 if values else 0, h)
    if st == _csx.EINVAL:
        raise IndexError("list index out of range")
    _csx.check(st, "csx_permute")
    return _result(h, lambda nnz: max(nnz, 1), A._pinned)


def cs_symperm(A, pinv, values):
    """Upper triangle of P A P' for a symmetric A whose upper triangle is stored (csparse.py:2220-2255)."""
    if not CS_CSC(A):
        return None
    room = _meta(A)[0]
    pv = None if pinv is None else _csx.i32(pinv)
    with _Resident(A) as dA:
        h = _csx.new_handle()
        st = _csx.lib().csx_symperm(dA.handle, _csx.pi(pv), 1 if values else 0, h)
    if st == _csx.EINVAL:
        raise IndexError("list index out of range")
    _csx.check(st, "csx_symperm")
    return _result(h, lambda nnz: max(room, 1), A._pinned)

def _plan(dT, kind):
    h = dT.plans.get(kind)
    if h is None:
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_tri_analyse(dT.handle, kind, h), "csx_tri_analyse")
        dT.plans[kind] = h
    return h


def _trisolve(T, x, kind):
    if not CS_CSC(T) or x is None:
        return False
    if not _meta(T)[1]:
        raise TypeError("'NoneType' object is not subscriptable")
    if T.m != T.n:
        raise IndexError("list index out of range")
    if _HOST_CHAINS[0] and not isinstance(x, dvec):
        # "tri.host_chains" (opt-in, cs_option): one host right-hand side on a chain-like factor -- the reference's loop on the
        # host inside libcsx, same bits (csx_tri_solve_list); anything else falls through to the device
        buf = _csx.f64(x[:T.n])
        taken = _csx.C.c_int(0)
        with _Resident(T) as dT:
            _csx.check(_csx.lib().csx_tri_solve_list(_plan(dT, kind), _csx.pd(buf), taken), "csx_tri_solve_list")
        if taken.value:
            x[:T.n] = buf.tolist() if isinstance(x, list) else buf
            return True
    dx, xhost = _vec_in(x, T.n, "x")
    nrhs = dx.k
    with _Resident(T) as dT:
        plan = _plan(dT, kind)
        _csx.check(_csx.lib().csx_tri_solve(plan, dx.handle, nrhs), "csx_tri_solve")
    _write_back(xhost, dx, T.n * nrhs)
    return True


_HOST_CHAINS = [False]


def cs_option(name, value):
    """csx_set_option from the drop-in module (DESIGN.md lists the names).  "tri.host_chains" = 1 additionally lets the list-level
    cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve / cs_cholsol / cholsol_factor(...).solve(list) try the host loop first."""
    _csx.check(_csx.lib().csx_set_option(name.encode(), int(value)), "csx_set_option")
    if name == "tri.host_chains":
        _HOST_CHAINS[0] = bool(value)


def _cholsol_list_on_host(plan, b, n):
    """cs_cholsol's solve sequence for a LIST b by csx_cholsol_solve_list ("tri.host_chains"); True when it was taken"""
    if not _HOST_CHAINS[0] or isinstance(b, dvec):
        return False
    buf = _csx.f64(b[:n])
    taken = _csx.C.c_int(0)
    _csx.check(_csx.lib().csx_cholsol_solve_list(plan, _csx.pd(buf), taken), "csx_cholsol_solve_list")
    if taken.value:
        b[:n] = buf.tolist() if isinstance(b, list) else buf
    return bool(taken.value)


def cs_lsolve(L, x):
    """Solve L x = b in place, diagonal first in each column (csparse.py:1330-1345)."""
    return _trisolve(L, x, TRI_L)


def cs_ltsolve(L, x):
    """Solve L' x = b in place (csparse.py:1348-1365)."""
    return _trisolve(L, x, TRI_LT)


def cs_usolve(U, x):
    """Solve U x = b in place, diagonal last in each column (csparse.py:2368-2385)."""
    return _trisolve(U, x, TRI_U)


def cs_utsolve(U, x):
    """Solve U' x = b in place (csparse.py:2460-2475)."""
    return _trisolve(U, x, TRI_UT)


def _perm_handle(p, n):
    if p is None:
        return _csx.H(0), None
    a = _csx.i32(p[:n])
    h = _csx.new_handle()
    _csx.check(_csx.lib().csx_ivec_upload(_csx.pi(a), n, h), "csx_ivec_upload")
    return h, h


def _permute(p, b, x, n, inverse):
    if x is None or b is None:
        return False
    if isinstance(b, dvec) and isinstance(x, dvec):
        h, tmp = _perm_handle(p, n)
        try:
            _csx.check(_csx.lib().csx_permute_vec(h, b.handle, x.handle, n, b.k, inverse), "csx_permute_vec")
        finally:
            _csx.free(tmp)
        return True
    # a single host list: the reference's own loop is the whole job
    if inverse:
        for k in range(n):
            x[p[k] if p is not None else k] = b[k]
    else:
        for k in range(n):
            x[k] = b[p[k] if p is not None else k]
    return True


def cs_ipvec(p, b, x, n):
    """x(p) = b (csparse.py:1264-1277); p None is the identity."""
    return _permute(p, b, x, n, 1)


def cs_pvec(p, b, x, n):
    """x = b(p) (csparse.py:1779-1792); p None is the identity."""
    return _permute(p, b, x, n, 0)


# ------------------------------------------------------------- Cholesky ----

def _pattern_np(A):
    """Column pointers and row indices of A as int32 arrays: straight from the device copy when there is one (a pinned
    or device-made matrix: that copy is the matrix), else from the lists."""
    n = A.n
    if A._dev is not None:
        m_, n_, nnz, hv = A._dev.info()
        p = np.empty(n + 1, dtype=np.int32)
        i = np.empty(max(nnz, 1), dtype=np.int32)
        _csx.check(_csx.lib().csx_csc_download(A._dev.handle, _csx.pi(p), _csx.pi(i), None), "csx_csc_download")
        return p, i[:nnz]
    p = _csx.i32(A.p[:n + 1])
    return p, _csx.i32(A.i[:int(p[n])])


def _amd_np(order, A):
    """Permutation (int32 array of n entries) for order 1 (A + A', square A), 2 (S'S with S = A less its dense rows) or
    3 (A'A), or None: a nested dissection of that graph (csx_order_nd_host)."""
    if not CS_CSC(A) or order not in (1, 2, 3):
        return None
    n = A.n
    if order == 1 and A.m == n:
        p, i = _pattern_np(A)                                   # the graph of A + A' is formed by the ordering itself
    else:
        # the pattern of A'A (csparse.py:236-256); order 2 first drops the dense rows of A (columns of A')
        Ap, Ai = _pattern_np(A)
        m = A.m
        P = cs_spalloc(m, n, max(len(Ai), 1), False, False)
        P.p, P.i, P.x = Ap.tolist(), Ai.tolist() if len(Ai) else [0], None
        AT = cs_transpose(P, False)
        if order == 2:
            dense = min(n - 2, max(16, int(10 * sqrt(n))))
            cs_fkeep(AT, lambda i, j, a, cnt: cnt[j] <= dense, np.diff(np.asarray(AT.p)).tolist())
        C = cs_multiply(AT, P)
        if C is None:
            return None
        p, i = _csx.i32(C.p[:n + 1]), _csx.i32(C.i[:C.p[n]])
    perm = np.empty(max(n, 1), dtype=np.int32)
    if _csx.load().csx_order_nd_host(n, _csx.pi(p), _csx.pi(i), _csx.pi(perm)) != _csx.OK:
        return None
    return perm[:n]


def cs_amd(order, A):
    """Fill-reducing ordering p (csparse.py:214-556): order 1 = for Cholesky / LU of a matrix with a symmetric pattern
    (graph of A + A'), 2 = for LU (graph of S'S, S = A without its dense rows), 3 = for QR (graph of A'A).
    The reference's implementation does not run, so there is no permutation to match: this is a nested
    dissection of the same graphs (breadth-first level separators, host C++), which gives the device a bushy
    elimination tree.  None for order 0 or bad input, like the reference."""
    perm = _amd_np(order, A)
    return None if perm is None else perm.tolist()


def cs_schol(order, A, _arrays=False):
    """Symbolic Cholesky analysis (csparse.py:2051-2072): ordering, etree, column counts.  order 0 =
    natural; order 1 = cs_amd (here a nested dissection).  The tree is built by host C++ inside libcsx,
    the column counts on the device when A is resident there.  (_arrays: internal -- S.parent / S.cp / S.pinv stay
    int32 arrays instead of becoming lists; cs_cholsol and cholsol_factor, which keep S to themselves, use it to skip
    half a dozen list conversions of length n.)"""
    if not CS_CSC(A) or order not in (0, 1):
        return None
    if order == 1:
        P = _amd_np(1, A)
        if P is None:
            return None
        pinv = np.empty(A.n, dtype=np.int32)
        pinv[P] = np.arange(A.n, dtype=np.int32)          # cs_pinv (csparse.py:1696-1708)
        C = cs_symperm(A, pinv, False)
        S = cs_schol(0, C, _arrays)
        if S is not None:
            S.pinv = pinv if _arrays else pinv.tolist()
        return S
    n = A.n
    parent = np.empty(max(n, 1), dtype=np.int32)
    cp = np.empty(n + 1, dtype=np.int32)
    if A._dev is not None and A.m == A.n:
        # device-resident matrix: tree on the host, column counts from the device's row-subtree walks
        st = _csx.lib().csx_schol(A._dev.handle, _csx.pi(parent), _csx.pi(cp))
    else:
        p = _csx.i32(A.p[:n + 1])
        i = _csx.i32(A.i[:int(p[n])])
        st = _csx.load().csx_schol_host(n, _csx.pi(p), _csx.pi(i), _csx.pi(parent), _csx.pi(cp))
    if st != _csx.OK:
        return None
    S = css()
    S.pinv = None
    S.q = None
    S.parent = parent[:n] if _arrays else parent[:n].tolist()
    S.cp = cp if _arrays else cp.tolist()
    S.unz = S.lnz = int(cp[n])
    return S


def cs_chol(A, S):
    """Numeric Cholesky L L' = P A P' (csparse.py:561-619); None if A is not
    positive definite.  The returned csn holds L as a device-backed `cs`."""
    if not CS_CSC(A) or S is None or S.cp is None or S.parent is None:
        return None
    n = A.n
    parent = _csx.i32(S.parent)
    cp = _csx.i32(S.cp)
    pinv = None if S.pinv is None else _csx.i32(S.pinv)
    with _Resident(A) as dA:
        h = _csx.new_handle()
        st = _csx.lib().csx_chol(dA.handle, _csx.pi(parent), _csx.pi(cp), _csx.pi(pinv), h)
    if st == _csx.ENOTSPD:
        return None
    _csx.check(st, "csx_chol")
    N = csn()
    N.L = _from_device(h, lambda nnz: max(nnz, 1))
    N.U = None
    N.pinv = None
    N.B = None
    return N


def spsolve_columns(G, B, pinv=None, lo=True, values=True):
    """cs_spsolve (csparse.py:2078-2113) for every column of B in one device call: a CSC matrix X whose column k
    lists the reach of B(:,k) in the reference's xi[top..n-1] order (cs_reach :1939-1958, cs_dfs :789-829) with the
    solution of G x = B(:,k) beside it, bit-identical to the reference called column by column.  G lower (lo) or
    upper triangular; pinv as in cs_lu (negative = no column yet).  values=False: the reaches alone.  None on bad
    input.  The list-level cs_spsolve / cs_reach / cs_dfs (one column, caller-owned work arrays) stay host functions."""
    if not CS_CSC(G) or not CS_CSC(B) or G.m != G.n or B.m != G.n:
        return None
    pv = None if pinv is None else _csx.i32(pinv[:G.n])
    with _Resident(G) as dG, _Resident(B) as dB:
        h = _csx.new_handle()
        st = _csx.lib().csx_spsolve(dG.handle, dB.handle, None if pv is None else _csx.pi(pv), 1 if lo else 0,
                                    1 if values else 0, h)
        if st == _csx.EINVAL:
            raise IndexError("list index out of range")
        _csx.check(st, "csx_spsolve")
    X = _from_device(h, lambda nnz: max(nnz, 1))
    if not (G._pinned and B._pinned):
        X._materialise()
        X._dev = None
        X._pinned = False
    return X


def reach_columns(G, B, pinv=None):
    """cs_reach (csparse.py:1939-1958) for every column of B: column k of the result = xi[top..n-1]."""
    return spsolve_columns(G, B, pinv, True, False)


def cs_updown(L, sigma, C, parent):
    """Sparse Cholesky rank-1 update (sigma = +1) / downdate (-1): L L' + sigma w w' with w = the one column of
    C, in place (csparse.py:2318-2365).  True on success; False on bad input or when the downdate is not positive
    definite (L is then changed exactly as far as the reference's loop gets).  L.x is bit-identical to the
    reference's.  A list-backed L is updated in its own list objects; a device-backed L stays on the device."""
    if not CS_CSC(L) or not CS_CSC(C) or parent is None:
        return False
    if not _meta(L)[1]:
        raise TypeError("'NoneType' object is not subscriptable")
    cnz = C.p[1] - C.p[0]
    if cnz <= 0:
        return True
    ci = _csx.i32(C.i[C.p[0]:C.p[1]])
    cx = _csx.f64(C.x[C.p[0]:C.p[1]])
    par = _csx.i32(parent[:L.n])
    ok = _csx.C.c_int(0)
    lazy = L._lazy
    with _Resident(L) as dL:
        for h in dL.plans.values():          # triangular-solve plans hold copies of the old values
            _csx.free(h)
        dL.plans.clear()
        dL.version += 1                      # cholsol_factor solvers built on this factor re-plan at their next solve
        st = _csx.lib().csx_updown(dL.handle, int(sigma), cnz, _csx.pi(ci), _csx.pd(cx), _csx.pi(par), ok)
        if st == _csx.EINVAL:
            raise IndexError("list index out of range")
        _csx.check(st, "csx_updown")
        if not lazy:                         # the caller holds L.x: update that list in place, like the reference
            m, n, nnz, hv = dL.info()
            x = np.empty(max(nnz, 1), dtype=np.float64)
            _csx.check(_csx.lib().csx_csc_download(dL.handle, None, None, _csx.pd(x)), "csx_csc_download")
            L._x[:nnz] = x[:nnz].tolist()
    return bool(ok.value)


def _solve_blocks_sharded(comm, b, nrhs, rows_in, rows_out, solve_block):
    """A batch of right-hand sides sharded by column block over the ranks of `comm` (independent units, no
    collective inside a block's solve).  The root (rank 0) passes b, a dvec rows_in-by-K block (or a list: K = 1); the
    other ranks pass None and nrhs = K.  Rank r gets columns [r k, (r + 1) k), k = ceil(K / world), of b
    (csx_block_cols + csx_comm_scatter_blocks), solve_block(block: dvec rows_in-by-k) returns its dvec rows_out-by-k
    solutions (the same object when the solve is in place), which return to the root (csx_comm_gather_blocks).
    Returns on the root the rows_out-by-K block (b itself, overwritten, when rows_in == rows_out; a new dvec otherwise)
    and the host list to write back to, if b was one; (None, None) elsewhere.  Every column has the bits of the
    unsharded solve: a sharded solve IS the unsharded one column by column."""
    lib = _csx.lib()
    root = comm.rank == 0
    K = comm.broadcast_object((b.k if isinstance(b, dvec) else 1) if root else None, 0)
    if nrhs is not None and nrhs != K:
        raise ValueError("solve: nrhs does not match the root's block")
    db = bhost = None
    if root:
        db, bhost = _vec_in(b, rows_in, "b")
    k = (K + comm.world - 1) // comm.world
    mine = dvec(rows_in, k)
    packed = dvec(rows_in * k * comm.world) if root else None     # world blocks of rows_in x k, rank order; pad columns = 0

    def _slot(buf, rows, r):
        h = _csx.new_handle()
        _csx.check(lib.csx_vec_wrap(_csx.C.c_void_p(buf.device_ptr() + 8 * rows * k * r), rows * k, h), "csx_vec_wrap")
        return h

    if root:
        for r in range(comm.world):
            c0 = r * k
            kk = max(0, min(k, K - c0))
            if kk <= 0:
                continue
            # columns [c0, c0 + kk) of B -> block r of `packed` (its first kk columns when the last block is short)
            tmp = dvec(rows_in, kk)
            _csx.check(lib.csx_block_cols(db.handle, rows_in, K, c0, kk, tmp.handle, 0), "csx_block_cols")
            slot = _slot(packed, rows_in, r)
            _csx.check(lib.csx_block_cols(slot, rows_in, k, 0, kk, tmp.handle, 1), "csx_block_cols")
            _csx.free(slot)
    comm.scatter_vec_blocks(packed.handle if root else None, mine.handle, rows_in * k, 0)
    sol = solve_block(mine)
    back = packed if rows_out == rows_in else (dvec(rows_out * k * comm.world) if root else None)
    comm.gather_vec_blocks(sol.handle, back.handle if root else None, rows_out * k, 0)
    if not root:
        return None, None
    out = db if rows_out == rows_in else dvec(rows_out, K)
    for r in range(comm.world):
        c0 = r * k
        kk = max(0, min(k, K - c0))
        if kk <= 0:
            continue
        slot = _slot(back, rows_out, r)
        tmp = dvec(rows_out, kk)
        _csx.check(lib.csx_block_cols(slot, rows_out, k, 0, kk, tmp.handle, 0), "csx_block_cols")
        _csx.check(lib.csx_block_cols(out.handle, rows_out, K, c0, kk, tmp.handle, 1), "csx_block_cols")
        _csx.free(slot)
    return out, bhost


def cs_cholsol(order, A, b):
    """Solve A x = b, A symmetric positive definite, upper triangle used; b is
    overwritten (csparse.py:622-644).  b may be a list (one system) or a dvec
    n-by-k block (k systems, factor once)."""
    if not CS_CSC(A) or b is None:
        return False
    n = A.n
    if order == 0 and A.m == n and _meta(A)[1]:
        # natural order: S = cs_schol, N = cs_chol and the solve plan in one library call, S never leaving the device
        # (csx_cholsol_factor); the reference's driver is always exact
        fused = _cholsol_factor_fused(A, True)
        if fused is None:
            return False
        L, plan = fused
        try:
            if _cholsol_list_on_host(plan, b, n):
                return True
            db, bhost = _vec_in(b, n, "b")
            _csx.check(_csx.lib().csx_cholsol_solve(plan, db.handle, db.k), "csx_cholsol_solve")
        finally:
            _csx.free(plan)
        _write_back(bhost, db, n * db.k)
        return True
    S = cs_schol(order, A, _arrays=True)
    N = cs_chol(A, S) if S is not None else None
    if S is None or N is None:
        return False
    pinv = None if S.pinv is None else _csx.i32(S.pinv)
    plan = _csx.new_handle()
    with _Resident(N.L) as dL:
        _csx.check(_csx.lib().csx_cholsol_plan(dL.handle, _csx.pi(pinv), plan), "csx_cholsol_plan")
        try:
            if _cholsol_list_on_host(plan, b, n):
                return True
            db, bhost = _vec_in(b, n, "b")
            _csx.check(_csx.lib().csx_cholsol_solve(plan, db.handle, db.k), "csx_cholsol_solve")
        finally:
            _csx.free(plan)
    _write_back(bhost, db, n * db.k)
    return True


def _cholsol_factor_fused(A, exact):
    """csx_cholsol_factor: (L as a device-backed cs, plan handle), or None when A is not positive definite."""
    hL, hP = _csx.new_handle(), _csx.new_handle()
    with _Resident(A) as dA:
        st = _csx.lib().csx_cholsol_factor(dA.handle, 1 if exact else 0, hL, hP)
    if st in (_csx.ENOTSPD, _csx.EINVAL):        # not positive definite; an index out of range (cs_schol gives None for it)
        return None
    _csx.check(st, "csx_cholsol_factor")
    return _from_device(hL, lambda nnz: max(nnz, 1)), hP


def _symbolic_of_factor(L):
    """cs_schol(0, A)'s result read off the factor: S.cp = L.p; S.parent[j] = the first row below the diagonal of column j of L
    (csparse.py:1136-1169 builds the same tree from A); as int32 arrays."""
    dev = L._dev
    m, n, nnz, hv = dev.info()
    p = np.empty(n + 1, dtype=np.int32)
    i = np.empty(max(nnz, 1), dtype=np.int32)
    _csx.check(_csx.lib().csx_csc_download(dev.handle, _csx.pi(p), _csx.pi(i), None), "csx_csc_download")
    parent = np.full(n, -1, dtype=np.int32)
    has = np.diff(p) > 1
    parent[has] = i[p[:-1][has] + 1]
    S = css()
    S.pinv = None
    S.q = None
    S.parent = parent
    S.cp = p
    S.unz = S.lnz = int(p[n])
    return S


def cholsol_factor(A, order=0, exact=None):
    """Factor once for many solves: returns a solver `solve(b)` where b is a list or a
    dvec n-by-k block (overwritten).  The batched form of cs_cholsol (csparse.py:622-644).
    The order of a solve's operations:
    exact=None (default): by the kind of right-hand side.  A LIST -- the reference's data model, the drop-in contract --
      is solved with the reference's operations in the reference's order: bit-identical to cs_lsolve + cs_ltsolve on
      the same L.  A dvec BLOCK -- the caller has left the reference's data model for the batched one -- is solved in the
      rounding-equal order, inside the 1e-10 that BASELINE.json's north_star grants x[]: dense blocks go to the matrix
      cores (blocked TRSM with explicit tile inverses, refused when an inverse is large), a big elimination tree to the
      supernodal schedule.  (G-spd, 128 right-hand sides: 2.4 ms against 4.8; bcsstk16: 0.4 ms against 6.6.)
    exact=True: every solve, blocks too, bit-identical to the reference's order.   exact=False: every solve rounding-equal.
    cs_cholsol, the reference's own driver, is always exact."""
    if not CS_CSC(A) or A.m != A.n:
        return None
    first_plan = None
    if order == 0 and A.m == A.n and _meta(A)[1]:
        # natural order: analysis, factorisation and plan in ONE library call, S never leaving the device (csx_cholsol_factor:
        # round 4's flow handed 40 MB of parent / cp to the host, back again, and re-read L twice to re-arrange it).  The plan
        # starts in the order blocks are solved in unless every solve is to be exact; a list switches it (a flag: the exact
        # kernel of a forest of equal blocks reads L.x itself).
        fused = _cholsol_factor_fused(A, exact is True)
        if fused is None:
            return None
        N = csn()
        N.L, first_plan = fused
        N.U, N.pinv, N.B = None, None, None
        S = None                                 # read off the factor when `symbolic` is asked for
        pinv = None
    else:
        S = cs_schol(order, A, _arrays=True)
        N = cs_chol(A, S) if S is not None else None
        if N is None:
            return None
        pinv = None if S.pinv is None else _csx.i32(S.pinv)
    # (the solver exposes S with lists, as cs_schol returns them -- made when `symbolic` is first read: at 5M columns the
    # three conversions take 0.3 s, sixty times the analysis itself)
    # The C plan BORROWS L's device arrays (CholPlan::L is not owned; the L' plan reads L.p / L.i / L.x directly),
    # so the factor must stay on the device for as long as the solver lives: N.L is pinned (reading F.L.p / .i / .x
    # copies to the host but keeps the device matrix), and the solver holds the _DevMatrix itself.
    cs_pin(N.L)
    dev = N.L._dev
    n = A.n
    start_exact = exact is True if first_plan is not None else exact is not False

    def _build():
        h = _csx.new_handle()
        _csx.check(_csx.lib().csx_cholsol_plan(dev.handle, _csx.pi(pinv), h), "csx_cholsol_plan")
        if not start_exact:
            _csx.check(_csx.lib().csx_cholsol_set_order(h, 0), "csx_cholsol_set_order")
        return h

    class _Solver(object):
        L = N.L

        @property
        def symbolic(self):
            nonlocal S
            if S is None:
                S = _symbolic_of_factor(N.L)
            for name in ("parent", "cp", "pinv"):
                v = getattr(S, name)
                if v is not None and not isinstance(v, list):
                    setattr(S, name, v.tolist())
            return S

        def __init__(self):
            self._dev = dev                      # keeps the device factor alive (see above)
            self._built = dev.version
            self.plan_handle = first_plan if first_plan is not None else _build()
            self._exact_now = start_exact            # the order the plan is in
            self._box = [self.plan_handle]
            self._fin = weakref.finalize(self, lambda box: _csx.free(box[0]), self._box)

        def _current(self):
            """The plan copies part of L's values (forward gather arrays, fragments): after cs_updown(F.L, ...) changed
            the factor in place it is rebuilt from the factor as it now stands."""
            if self._built != dev.version:
                _csx.free(self.plan_handle)
                self.plan_handle = self._box[0] = _build()
                self._exact_now = start_exact
                self._built = dev.version
            return self.plan_handle

        def _plan_for(self, block):
            """The plan in the order this right-hand side is solved in (see cholsol_factor): switching is a flag; the
            rounding-equal order's operands (tile inverses, supernodal schedule) are built the first time it is asked for."""
            h = self._current()
            want_exact = exact if exact is not None else not block
            if want_exact != self._exact_now:
                _csx.check(_csx.lib().csx_cholsol_set_order(h, 1 if want_exact else 0), "csx_cholsol_set_order")
                self._exact_now = want_exact
            return h

        def info(self):
            a, b, c = _csx.C.c_int32(), _csx.C.c_int32(), _csx.C.c_int32()
            _csx.check(_csx.lib().csx_cholsol_info(self._current(), a, b, c), "csx_cholsol_info")
            return {"fused_local": a.value in (1, 2, 3, 5), "dense_block": c.value if a.value in (2, 3) else 0,  # dense kernels in use
                    "matrix_cores": a.value in (3, 5), "trees": b.value, "max_nodes": c.value}

        def solve(self, b, comm=None, nrhs=None):
            """b: a list (one system) or a dvec n-by-k block, overwritten with the solutions.
            comm (a shard.Comm of more than one rank, every rank holding this same factor -- factored redundantly or
            shipped with comm.bcast_csc): the batch is sharded by right-hand-side block.  The root (rank 0)
            passes the n-by-K block, the other ranks pass None and nrhs = K; rank r solves columns [r k, (r + 1) k),
            k = ceil(K / world): blocks leave the root (csx_comm_scatter_blocks), every rank runs the sequence of
            csparse.py:640-643 on its block with no communication, the solutions return (csx_comm_gather_blocks) and
            the root's block is overwritten.  Every column has the bits of the unsharded solve."""
            if comm is not None and comm.world > 1:
                return self._solve_sharded(b, comm, nrhs)
            if not isinstance(b, dvec) and (exact is None or exact) and _cholsol_list_on_host(self._plan_for(False), b, n):
                return True
            db, bhost = _vec_in(b, n, "b")
            _csx.check(_csx.lib().csx_cholsol_solve(self._plan_for(isinstance(b, dvec)), db.handle, db.k), "csx_cholsol_solve")
            _write_back(bhost, db, n * db.k)
            return True

        def _solve_sharded(self, b, comm, nrhs):
            # every rank solves in the order the ROOT's right-hand side asks for
            plan = self._plan_for(comm.broadcast_object(isinstance(b, dvec) if comm.rank == 0 else None, 0))

            def block(mine):
                _csx.check(_csx.lib().csx_cholsol_solve(plan, mine.handle, mine.k), "csx_cholsol_solve")
                return mine

            out, bhost = _solve_blocks_sharded(comm, b, nrhs, n, n, block)
            if out is not None:
                _write_back(bhost, out, n * out.k)
            return True

    return _Solver()


# -------------------------------------------------------------------- LU ----

def _cs_from_arrays(m, n, p, i, x):
    C = cs_spalloc(m, n, len(i), True, False)
    C.p, C.i, C.x = p, i if i else [0], x if x else [0.0]
    C.nzmax = max(len(i), 1) if i else 0
    return C


def cs_lu(A, S, tol):
    """Sparse LU with threshold partial pivoting, P A = L U (csparse.py:1370-1451).
    Host C++ (csx_lu_host): pivot search is serial and data dependent.  L has its unit
    diagonal first in every column, U its diagonal last -- what cs_lsolve / cs_usolve need."""
    if not CS_CSC(A) or S is None:
        return None
    if not _meta(A)[1]:        # asked of the device for a device-made A: no download, the device copy stays
        raise TypeError("'NoneType' object is not subscriptable")
    n = A.n
    if A.m != n:
        raise IndexError("list index out of range")
    if S.q is not None:
        # column k of the factorisation is column q[k] of A (csparse.py:1405): the same as factoring A Q in natural order
        Sq = css()
        Sq.q, Sq.pinv, Sq.lnz, Sq.unz = None, None, S.lnz, S.unz
        return cs_lu(cs_permute(A, None, S.q, True), Sq, tol)
    if n >= 4096 or A._dev is not None:
        # a batch of small independent blocks (block-diagonal up to a symmetric permutation) factors on the device,
        # one workgroup per block; anything else comes back with done = 0 and takes the host code below
        pinv = np.empty(max(n, 1), dtype=np.int32)
        hL, hU, done = _csx.new_handle(), _csx.new_handle(), _csx.C.c_int(0)
        with _Resident(A) as dA:             # ONE residency for both device attempts (an unpinned A is uploaded once)
            st = _csx.lib().csx_lu_blocks(dA.handle, float(tol), hL, hU, _csx.pi(pinv), done)
            if st == _csx.ENOTSPD:
                return None
            _csx.check(st, "csx_lu_blocks")
            if not done.value:
                # one connected matrix: columns scheduled by the column elimination tree, a lane per column (csx_lu_etree);
                # done = 0 again for anything but a shallow tree with short columns, which stays with the host loop
                st = _csx.lib().csx_lu_etree(dA.handle, float(tol), hL, hU, _csx.pi(pinv), done)
                if st == _csx.ENOTSPD:
                    return None
                _csx.check(st, "csx_lu_etree")
        if done.value:
            N = csn()
            N.L = _from_device(hL, lambda nnz: max(nnz, 1))
            N.U = _from_device(hU, lambda nnz: max(nnz, 1))
            N.pinv = pinv[:n].tolist()
            N.B = None
            return N
    p = _csx.i32(A.p[:n + 1])
    nnz = int(p[n])
    i, x = _csx.i32(A.i[:nnz]), _csx.f64(A.x[:nnz])
    C = _csx.C
    out = [C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)(),
           C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()]
    pinv = np.empty(max(n, 1), dtype=np.int32)
    lib = _csx.load()
    st = lib.csx_lu_host(n, _csx.pi(p), _csx.pi(i), _csx.pd(x), float(tol), *[C.byref(o) for o in out], _csx.pi(pinv))
    if st == _csx.ENOTSPD:
        return None
    _csx.check(st, "csx_lu_host")
    try:
        Lp = np.ctypeslib.as_array(out[0], shape=(n + 1,)).tolist()
        Up = np.ctypeslib.as_array(out[3], shape=(n + 1,)).tolist()
        lnz, unz = Lp[n], Up[n]
        Li = np.ctypeslib.as_array(out[1], shape=(max(lnz, 1),))[:lnz].tolist()
        Lx = np.ctypeslib.as_array(out[2], shape=(max(lnz, 1),))[:lnz].tolist()
        Ui = np.ctypeslib.as_array(out[4], shape=(max(unz, 1),))[:unz].tolist()
        Ux = np.ctypeslib.as_array(out[5], shape=(max(unz, 1),))[:unz].tolist()
    finally:
        for o in out:
            lib.csx_host_free(C.cast(o, C.c_void_p))
    N = csn()
    N.L = _cs_from_arrays(n, n, Lp, Li, Lx)
    N.U = _cs_from_arrays(n, n, Up, Ui, Ux)
    N.pinv = pinv[:n].tolist()
    N.B = None
    return N


def cs_lusol(order, A, b, tol):
    """Solve A x = b by LU; b is overwritten (csparse.py:1456-1478): host factorisation, then
    x = b(p); L\\x; U\\x; b(q) = x with the triangular solves on the device."""
    if not CS_CSC(A) or b is None:
        return False
    n = A.n
    S = cs_sqr(order, A, False)
    N = cs_lu(A, S, tol) if S is not None else None
    if S is None or N is None:
        return False
    x = dvec(n, b.k) if isinstance(b, dvec) else xalloc(n)      # b: a list (one system) or a dvec n-by-k block (k systems)
    cs_ipvec(N.pinv, b, x, n)
    cs_lsolve(N.L, x)
    cs_usolve(N.U, x)
    cs_ipvec(S.q, x, b, n)
    return True


def lusol_factor(A, order=0, tol=1.0, exact=None):
    """Factor once for many solves -- the batched form of cs_lusol (csparse.py:1456-1478): cs_sqr + cs_lu once, then
    solve(b) runs the reference's sequence x = b(p); L \\ x; U \\ x; b(q) = x (:1474-1477) on the device for a list (one
    system) or a dvec n-by-k block (k systems, overwritten): csx_permute_vec, csx_tri_solve on L and on U,
    csx_permute_vec.  The order of a solve's operations follows cholsol_factor's rule: exact=None (default): a LIST is
    solved in the reference's order, bit-identical to cs_lusol; a dvec BLOCK in the rounding-equal order (x[] within the
    1e-10 of BASELINE.json's north_star) -- which differs from the exact one only where the factors fall into many small
    independent components of at most 80 rows (config 3's W: 1 493 of 67), solved densely on the matrix cores then
    (csx_tri_set_order; refused when an inverse of a diagonal tile is large); exact=True: every solve bit-identical to
    cs_lusol on that column; exact=False: every solve rounding-equal.  cs_lusol, the reference's driver, is always exact.
    solve(b, comm=..., nrhs=K) shards the block by right-hand-side block over the ranks of a shard.Comm, every rank
    holding this factor; see cholsol_factor.  None when A is not square CSC or singular."""
    if not CS_CSC(A) or A.m != A.n:
        return None
    S = cs_sqr(order, A, False)
    N = cs_lu(A, S, tol) if S is not None else None
    if N is None:
        return None
    n = A.n
    L, U = cs_pin(N.L), cs_pin(N.U)
    hp, keep_p = _perm_handle(N.pinv, n)
    hq, keep_q = _perm_handle(S.q, n)

    class _Solver(object):
        factors, symbolic = N, S

        def __init__(self):
            self._fin = weakref.finalize(self, lambda hs: [_csx.free(h) for h in hs if h is not None], [keep_p, keep_q])

        def _block(self, blk, in_exact_order=True):
            # x(pinv) = b, L x = x, U x = x, b(q) = x (csparse.py:1470-1473) as ONE library call: in the rounding-equal order on
            # forests of small components the two permutations ride on the two sweeps (csx_lusol_solve)
            lib = _csx.lib()
            x = dvec(n, blk.k)
            with _Resident(L) as dL, _Resident(U) as dU:
                pl, pu = _plan(dL, TRI_L), _plan(dU, TRI_U)      # (shared with the list-level cs_lsolve / cs_usolve on this factor: the order is set per solve)
                fused = _csx.C.c_int(0)
                try:
                    for plan in (pl, pu):
                        _csx.check(lib.csx_tri_set_order(plan, 1 if in_exact_order else 0), "csx_tri_set_order")
                    _csx.check(lib.csx_lusol_solve(pl, pu, hp, hq, blk.handle, x.handle, blk.k, _csx.C.byref(fused)), "csx_lusol_solve")
                finally:
                    lib.csx_tri_set_order(pl, 1)
                    lib.csx_tri_set_order(pu, 1)
            self.last_fused = bool(fused.value)
            return blk

        def info(self):
            """which of the two triangular solves run on the matrix cores in the rounding-equal order, and the guard's measure"""
            out = {}
            for name, M, kind in (("L", L, TRI_L), ("U", U, TRI_U)):
                with _Resident(M) as dM:
                    mc, g = _csx.C.c_int32(0), _csx.C.c_double(0.0)
                    plan = _plan(dM, kind)
                    _csx.check(_csx.lib().csx_tri_set_order(plan, 0), "csx_tri_set_order")
                    _csx.check(_csx.lib().csx_tri_order_info(plan, mc, g), "csx_tri_order_info")
                    _csx.lib().csx_tri_set_order(plan, 1)
                    out[name] = {"matrix_cores": bool(mc.value), "growth": g.value}
            return out

        def solve(self, b, comm=None, nrhs=None):
            if comm is not None and comm.world > 1:
                # every rank solves in the order the ROOT's right-hand side asks for
                ex = comm.broadcast_object((exact if exact is not None else not isinstance(b, dvec)) if comm.rank == 0 else None, 0)
                out, bhost = _solve_blocks_sharded(comm, b, nrhs, n, n, lambda blk: self._block(blk, ex))
                if out is not None:
                    _write_back(bhost, out, n * out.k)
                return True
            db, bhost = _vec_in(b, n, "b")
            self._block(db, exact if exact is not None else not isinstance(b, dvec))
            _write_back(bhost, db, n * db.k)
            return True

    return _Solver()


# -------------------------------------------------------------------- QR ----
# cs_qr: host C++ (csx_qr_host), on the device for batches of small independent blocks (csx_qr_blocks); the Q' x step of
# the solve (cs_happly for every reflection) and the triangular solves on the device.

def cs_etree(A, ata):
    """Elimination tree of A (ata False; upper triangle used) or of A'A (csparse.py:1136-1169)."""
    if not CS_CSC(A):
        return None
    m, n = A.m, A.n
    Ap, Ai = A.p, A.i
    parent = [-1] * n
    anc = [-1] * n
    prev = [-1] * m if ata else None
    for k in range(n):
        for p in range(Ap[k], Ap[k + 1]):
            i = prev[Ai[p]] if ata else Ai[p]
            while i != -1 and i < k:
                up = anc[i]
                anc[i] = k
                if up == -1:
                    parent[i] = k
                i = up
            if ata:
                prev[Ai[p]] = k
    return parent


def cs_post(parent, n):
    """Postorder of a forest (csparse.py:1711-1742, :2258-2289)."""
    if parent is None:
        return None
    first_child = [-1] * n
    sibling = [-1] * n
    for j in range(n - 1, -1, -1):
        if parent[j] != -1:
            sibling[j] = first_child[parent[j]]
            first_child[parent[j]] = j
    post = []
    for root in range(n):
        if parent[root] != -1:
            continue
        stack = [root]
        while stack:
            c = first_child[stack[-1]]
            if c == -1:
                post.append(stack.pop())
            else:
                first_child[stack[-1]] = sibling[c]
                stack.append(c)
    return post


def cs_counts(A, parent, post, ata):
    """Column counts of L L' = A (ata False: the upper triangle of a square A) or L L' = A'A (ata True), given the elimination
    tree and its postorder (csparse.py:703-764; cs_leaf :1280-1304, _init_ata :677-700).  Host C++ (csx_counts_host): a symbolic
    step in front of the hot path.  None on bad input, like the reference."""
    if not CS_CSC(A) or parent is None or post is None:
        return None
    m, n = A.m, A.n
    if not ata and m != n:
        return None
    p = _csx.i32(A.p[:n + 1])
    i = _csx.i32(A.i[:int(p[n])])
    par, po = _csx.i32(parent[:n]), _csx.i32(post[:n])
    if len(par) < n or len(po) < n:
        return None
    cnt = np.empty(max(n, 1), dtype=np.int32)
    st = _csx.load().csx_counts_host(m, n, _csx.pi(p), _csx.pi(i), _csx.pi(par), _csx.pi(po), 1 if ata else 0, _csx.pi(cnt))
    if st != _csx.OK:
        return None
    return cnt[:n].tolist()


def cs_sqr(order, A, qr):
    """Symbolic ordering and analysis for QR or LU (csparse.py:2187-2217).  order 0 natural, 1 / 2 / 3 as cs_amd.
    LU: only the column ordering S.q and the reference's size guesses.  QR: the column elimination tree, the column
    counts of R and cs_vcount (leftmost, pinv, m2, entries of V) of A Q, in host C++ (csx_sqr_host)."""
    if not CS_CSC(A) or order not in (0, 1, 2, 3):
        return None
    n = A.n
    S = css()
    S.q = cs_amd(order, A)
    if order and S.q is None:
        return None
    S.pinv = None
    if not qr:
        S.unz = S.lnz = 4 * _meta(A)[0] + n
        return S
    m = A.m
    C = cs_permute(A, None, S.q, False) if order else A
    Ap = _csx.i32(C.p[:n + 1])
    nnz = int(Ap[n])
    Ai = _csx.i32(C.i[:nnz]) if nnz else np.zeros(1, np.int32)
    parent, cp = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
    pinv, leftmost = np.empty(max(m + n, 1), np.int32), np.empty(max(m, 1), np.int32)
    m2, vnz, rnz = _csx.C.c_int32(0), _csx.C.c_int64(0), _csx.C.c_int64(0)
    st = _csx.load().csx_sqr_host(m, n, _csx.pi(Ap), _csx.pi(Ai), _csx.pi(parent), _csx.pi(cp), _csx.pi(pinv),
                                  _csx.pi(leftmost), m2, vnz, rnz)
    if st == _csx.EINVAL:
        raise IndexError("list index out of range")
    _csx.check(st, "csx_sqr_host")
    S.parent, S.cp = parent[:n].tolist(), cp[:n].tolist()
    S.pinv, S.leftmost = pinv[:m + n].tolist(), leftmost[:m].tolist()
    S.m2, S.lnz, S.unz = int(m2.value), int(vnz.value), int(rnz.value)
    return S


def cs_house(x, x_offset, beta, n):
    """Householder reflection (I - beta v v') x = s e1; x is overwritten with v (csparse.py:1238-1261)."""
    if x is None or beta is None:
        return -1
    sigma = 0
    for i in range(1, n):
        sigma += x[x_offset + i] * x[x_offset + i]
    x0 = x[x_offset]
    if sigma == 0:
        s = abs(x0)
        beta[0] = 2.0 if x0 <= 0 else 0.0
        x[x_offset] = 1
    else:
        s = sqrt(x0 * x0 + sigma)
        x[x_offset] = x0 - s if x0 <= 0 else -sigma / (x0 + s)
        beta[0] = -1.0 / (s * x[x_offset])
    return s


def cs_happly(V, i, beta, x):
    """x = (I - beta v v') x with v = V(:, i) (csparse.py:1216-1235)."""
    if not CS_CSC(V) or x is None:
        return False
    Vi, Vx = V.i, V.x
    lo, hi = V.p[i], V.p[i + 1]
    tau = 0
    for p in range(lo, hi):
        tau += Vx[p] * x[Vi[p]]
    tau *= beta
    for p in range(lo, hi):
        x[Vi[p]] -= Vx[p] * tau
    return True


def cs_qr(A, S):
    """Sparse Householder QR, A = Q R (csparse.py:1797-1870).  N.L = V, N.U = R (diagonal last in
    every column), N.B = beta.  Host C++ (csx_qr_host, csx_host.cpp): column by column along the column
    elimination tree, like cs_lu a sequence of data-dependent steps -- except for a square matrix that is a batch of
    small independent blocks, which factors on the device (csx_qr_blocks: one lane per block, same results bit for bit)."""
    if not CS_CSC(A) or S is None:
        return None
    if not _meta(A)[1]:
        raise TypeError("'NoneType' object is not subscriptable")
    m, n, m2 = A.m, A.n, S.m2
    q = None if S.q is None else _csx.i32(S.q)
    parent, pinv, leftmost = _csx.i32(S.parent), _csx.i32(S.pinv), _csx.i32(S.leftmost)
    if q is None and m == n and m2 == m and (n >= 4096 or A._dev is not None):
        # a batch of small independent blocks factors on the device, one lane per block running the host code's loop
        # (csx_qr_blocks); anything else comes back with done = 0 and takes the host code below
        beta = np.zeros(max(n, 1))
        hV, hR, done = _csx.new_handle(), _csx.new_handle(), _csx.C.c_int(0)
        with _Resident(A) as dA:
            st = _csx.lib().csx_qr_blocks(dA.handle, _csx.pi(parent), _csx.pi(pinv), _csx.pi(leftmost), m2, hV, hR,
                                          _csx.pd(beta), done)
        if st == _csx.EINVAL:
            raise IndexError("list index out of range")
        _csx.check(st, "csx_qr_blocks")
        if done.value:
            N = csn()
            N.L = _from_device(hV, lambda nnz: max(nnz, 1))
            N.U = _from_device(hR, lambda nnz: max(nnz, 1))
            N.B = beta[:n].tolist()
            N.pinv = None
            return N
    Ap = _csx.i32(A.p[:n + 1])
    nnz = int(Ap[n])
    Ai, Ax = _csx.i32(A.i[:nnz]), _csx.f64(A.x[:nnz])
    vcap, rcap = max(int(S.lnz), 1), max(int(S.unz), 1)
    Vp, Rp = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    Vi, Ri = np.zeros(vcap, np.int32), np.zeros(rcap, np.int32)
    Vx, Rx, beta = np.zeros(vcap), np.zeros(rcap), np.zeros(max(n, 1))
    st = _csx.load().csx_qr_host(m, n, m2, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), _csx.pi(q), _csx.pi(parent), _csx.pi(pinv),
                                 _csx.pi(leftmost), vcap, rcap, _csx.pi(Vp), _csx.pi(Vi), _csx.pd(Vx), _csx.pi(Rp),
                                 _csx.pi(Ri), _csx.pd(Rx), _csx.pd(beta))
    if st == _csx.EINVAL:
        raise IndexError("list index out of range")
    _csx.check(st, "csx_qr_host")
    N = csn()
    N.L = V = cs_spalloc(m2, n, vcap, True, False)
    N.U = R = cs_spalloc(m2, n, rcap, True, False)
    V.p, V.i, V.x = Vp.tolist(), Vi.tolist(), Vx.tolist()
    R.p, R.i, R.x = Rp.tolist(), Ri.tolist(), Rx.tolist()
    N.B = beta[:n].tolist()
    N.pinv = None
    return N


def _apply_q(N, x, transpose):
    """x <- Q' x (transpose) or Q x for the Householder vectors in N.L / N.B (csx_qr_apply_host)."""
    V = N.L
    n = V.n
    Vp = _csx.i32(V.p[:n + 1])
    vnz = int(Vp[n])
    xv = _csx.f64(x)
    _csx.check(_csx.load().csx_qr_apply_host(n, _csx.pi(Vp), _csx.pi(_csx.i32(V.i[:vnz])), _csx.pd(_csx.f64(V.x[:vnz])),
                                             _csx.pd(_csx.f64(N.B)), 1 if transpose else 0, _csx.pd(xv)), "csx_qr_apply_host")
    x[:] = xv.tolist()


def _square_view(T):
    """R from cs_qr is m2-by-n with nothing below row n: present it as n-by-n to the device solves."""
    if T.m == T.n:
        return T
    V = cs()
    V.m = V.n = T.n
    V.nz, V.nzmax = -1, T.nzmax
    V.p, V.i, V.x = T.p, T.i, T.x
    return V


def cs_qrsol(order, A, b):
    """Least squares (m >= n) or minimum-norm solution (m < n) by QR; b (size max(m, n)) is
    overwritten with x (csparse.py:1875-1912).  The factorisation and the Householder applications
    run on the host, the triangular solves cs_usolve / cs_utsolve on the device."""
    if not CS_CSC(A) or b is None:
        return False
    n, m = A.n, A.m
    if m >= n:
        S = cs_sqr(order, A, True)
        N = cs_qr(A, S) if S is not None else None
        if S is None or N is None:
            return False
        x = xalloc(S.m2)
        cs_ipvec(S.pinv, b, x, m)
        _apply_q(N, x, True)
        cs_usolve(_square_view(N.U), x)
        cs_ipvec(S.q, x, b, n)
    else:
        AT = cs_transpose(A, True)
        S = cs_sqr(order, AT, True)
        N = cs_qr(AT, S) if S is not None else None
        if AT is None or S is None or N is None:
            return False
        x = xalloc(S.m2)
        cs_pvec(S.q, b, x, m)
        cs_utsolve(_square_view(N.U), x)
        _apply_q(N, x, False)
        cs_pvec(S.pinv, x, b, n)
    return True


def apply_q(N, X, transpose=True):
    """X <- Q' X (transpose) or Q X for the Householder vectors of N = cs_qr(A, S), X a dvec block of N.L.m rows
    (cs_happly, csparse.py:1216-1235, for every reflection and every column of X, on the device: csx_happly)."""
    if not isinstance(X, dvec) or X.n < N.L.m:
        raise IndexError("list index out of range")
    beta = dvec(np.asarray(N.B, dtype=np.float64) if len(N.B) else np.zeros(1))
    with _Resident(N.L) as dV:
        _csx.check(_csx.lib().csx_happly(dV.handle, beta.handle, X.handle, X.k, 1 if transpose else 0), "csx_happly")
    return True


def qrsol_factor(A, order=0):
    """Factor once (cs_sqr + cs_qr on the host), solve least-squares problems min ||A x - b|| for blocks of right-hand
    sides on the device: the solve sequence of cs_qrsol for m >= n (csparse.py:1893-1898) -- x = P b, Q' x, solve R x,
    x(q) -- as csx_permute_vec, csx_happly, csx_tri_solve.  solve(B): B a dvec m-by-k block or a list of m entries;
    returns the n-by-k solutions as a new dvec (or overwrites the list's first n entries, like cs_qrsol).  Every column
    is bit-identical to cs_qrsol on that column."""
    if not CS_CSC(A) or A.m < A.n:
        return None
    S = cs_sqr(order, A, True)
    N = cs_qr(A, S) if S is not None else None
    if N is None:
        return None
    m, n, m2 = A.m, A.n, S.m2
    R = cs_pin(_square_view(N.U))
    cs_pin(N.L)

    class _Solver(object):
        factors, symbolic = N, S

        def _block(self, db):
            X = dvec(m2, db.k)                       # zeros: the fictitious rows stay zero
            cs_ipvec(S.pinv, db, X, m)               # x(pinv) = b
            apply_q(N, X, True)
            cs_usolve(R, X)                          # the first n rows of the block
            out = dvec(n, db.k)
            cs_ipvec(S.q, X, out, n)
            return out

        def solve(self, b, comm=None, nrhs=None):
            """comm (a shard.Comm of more than one rank, every rank holding these factors): the block is sharded by
            right-hand-side block (see cholsol_factor) -- the root passes the m-by-K block and gets the n-by-K
            solutions, the other ranks pass None and nrhs = K and get None."""
            if comm is not None and comm.world > 1:
                host = b if (comm.rank == 0 and not isinstance(b, dvec)) else None
                src = dvec(np.asarray(b[:m], dtype=np.float64)) if host is not None else b
                out, _ = _solve_blocks_sharded(comm, src, nrhs, m, n, self._block)
                if host is not None:
                    host[:n] = out.numpy().reshape(-1)[:n].tolist()
                    return True
                return out
            host = None if isinstance(b, dvec) else b
            db = b if isinstance(b, dvec) else dvec(np.asarray(b[:m], dtype=np.float64))
            if db.n < m:
                raise IndexError("list index out of range")
            out = self._block(db)
            if host is not None:
                host[:n] = out.numpy().reshape(-1)[:n].tolist()
                return True
            return out

    return _Solver()


def device_name():
    return _csx.device_info()
