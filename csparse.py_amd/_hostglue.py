"""Host-side helpers of the reference that sit INSIDE its factorisations: the sparse
right-hand-side triangular solve and the reachability searches that feed it, and the row-pattern search of
the up-looking Cholesky.  In this build cs_lu and cs_chol do that work in C++ / on the device, so these are
provided for callers of the reference's own names only; plain Python on lists, the reference's argument
conventions (offsets into shared work arrays, marks kept as sign flips of G.p / w and undone on return).

cs_dfs      csparse.py:789-829      cs_reach    csparse.py:1939-1958
cs_spsolve  csparse.py:2078-2113    cs_ereach   csparse.py:1094-1131
"""


def CS_FLIP(i):
    return -i - 2


def CS_UNFLIP(i):
    return CS_FLIP(i) if i < 0 else i


def CS_MARKED(w, j):
    return w[j] < 0


def CS_MARK(w, j):
    w[j] = CS_FLIP(w[j])


def _is_csc(A):
    return A is not None and getattr(A, "nz", 0) == -1


def cs_dfs(j, G, top, xi, xi_offset, pstack, pstack_offset, pinv, pinv_offset=0):
    """Depth-first search from node j of the graph of G (columns renamed through pinv); finished nodes
    are pushed on the stack that grows downwards from xi[xi_offset + top].  Returns the new top."""
    if not _is_csc(G) or xi is None or pstack is None:
        return -1
    Gp, Gi = G.p, G.i
    depth = 0
    xi[xi_offset] = j
    while depth >= 0:
        node = xi[xi_offset + depth]
        col = pinv[pinv_offset + node] if pinv is not None else node
        if not CS_MARKED(Gp, node):
            CS_MARK(Gp, node)                                   # first visit: remember where its scan starts
            pstack[pstack_offset + depth] = 0 if col < 0 else CS_UNFLIP(Gp[col])
        stop = 0 if col < 0 else CS_UNFLIP(Gp[col + 1])
        descended = False
        p = pstack[pstack_offset + depth]
        while p < stop:
            nxt = Gi[p]
            if not CS_MARKED(Gp, nxt):
                pstack[pstack_offset + depth] = p              # resume here when we come back
                depth += 1
                xi[xi_offset + depth] = nxt
                descended = True
                break
            p += 1
        if not descended:
            depth -= 1
            top -= 1
            xi[xi_offset + top] = node
    return top


def cs_reach(G, B, k, xi, pinv):
    """xi[top..n-1] = nodes reachable from the pattern of B(:,k) in the graph of G, in topological order;
    xi[n..2n-1] is work space.  G.p is restored.  Returns top, -1 on bad input."""
    if not _is_csc(G) or not _is_csc(B) or xi is None:
        return -1
    n, Gp = G.n, G.p
    top = n
    for p in range(B.p[k], B.p[k + 1]):
        if not CS_MARKED(Gp, B.i[p]):
            top = cs_dfs(B.i[p], G, top, xi, 0, xi, n, pinv, 0)
    for p in range(top, n):
        CS_MARK(Gp, xi[p])
    return top


def cs_spsolve(G, B, k, xi, x, pinv, lo):
    """Solve G x = B(:,k) for a lower (lo true: diagonal first) or upper (diagonal last) triangular G
    and a sparse right-hand side; x is dense, only its entries xi[top..n-1] are meaningful."""
    if not _is_csc(G) or not _is_csc(B) or xi is None or x is None:
        return -1
    Gp, Gi, Gx, n = G.p, G.i, G.x, G.n
    top = cs_reach(G, B, k, xi, pinv)
    for p in range(top, n):
        x[xi[p]] = 0
    for p in range(B.p[k], B.p[k + 1]):
        x[B.i[p]] = B.x[p]
    for px in range(top, n):
        j = xi[px]
        col = pinv[j] if pinv is not None else j
        if col < 0:
            continue                                            # a row that is not yet pivotal
        first, last = Gp[col], Gp[col + 1]
        x[j] /= Gx[first if lo else last - 1]
        xj = x[j]
        for p in (range(first + 1, last) if lo else range(first, last - 1)):
            x[Gi[p]] -= Gx[p] * xj
    return top


def cs_ereach(A, k, parent, s, s_offset, w):
    """Pattern of row k of the Cholesky factor: s[s_offset + top .. s_offset + n - 1], in an order in
    which the up-looking solve may process it.  w must be non-negative on entry and is restored."""
    if not _is_csc(A) or parent is None or s is None or w is None:
        return -1
    top = n = A.n
    CS_MARK(w, k)
    for p in range(A.p[k], A.p[k + 1]):
        i = A.i[p]
        if i > k:
            continue                                            # only the upper triangle takes part
        path = 0
        while not CS_MARKED(w, i):                              # climb the tree until a marked node
            s[s_offset + path] = i
            path += 1
            CS_MARK(w, i)
            i = parent[i]
        while path > 0:                                         # push the path, root end first
            path -= 1
            top -= 1
            s[s_offset + top] = s[s_offset + path]
    for p in range(top, n):
        CS_MARK(w, s[s_offset + p])
    CS_MARK(w, k)
    return top
