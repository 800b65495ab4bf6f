/* CPU oracle, plain C  --  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Restates the hot-path algorithms of the reference (/root/reference/csparse.py,
 * cited as csparse.py:N) on int32 index / float64 value arrays, keeping the
 * reference's order of floating-point operations (build with -ffp-contract=off:
 * one rounding per multiply and per add, like CPython floats).  It exists so
 * parity tests and bench.py's cpu_baseline leg can run at sizes the list-based
 * Python oracle (oracle/csparse_oracle.py) cannot reach in seconds.
 *
 * Pin: tests/test_oracle_golden.py::test_c_oracle_* compares every function
 * here bit-for-bit with the Python oracle, which is itself pinned to vectors
 * produced by the unmodified reference (oracle/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Status codes: 0 ok, 1 bad argument, 2 zero pivot
 * (Python raises ZeroDivisionError there), 3 not positive definite.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- csparse.py:1199-1213 -------------------------------------------- */
int co_gaxpy(int m, int n, const int *Ap, const int *Ai, const double *Ax,
             const double *x, double *y)
{
    (void)m;
    if (!Ap || !Ai || !Ax || !x || !y) return 1;
    for (int j = 0; j < n; j++) {
        double xj = x[j];
        for (int p = Ap[j]; p < Ap[j + 1]; p++) {
            double t = Ax[p] * xj;
            y[Ai[p]] = y[Ai[p]] + t;
        }
    }
    return 0;
}

/* ---- csparse.py:767-784 ---------------------------------------------- */
long long co_cumsum(int *p, int *c, int n)
{
    long long total = 0;
    if (!p || !c) return -1;
    for (int k = 0; k < n; k++) {
        p[k] = (int)total;
        total += c[k];
        c[k] = p[k];
    }
    p[n] = (int)total;
    return total;
}

/* ---- csparse.py:2292-2315: C is n-by-m, Cp has m+1 slots -------------- */
int co_transpose(int m, int n, const int *Ap, const int *Ai, const double *Ax,
                 int *Cp, int *Ci, double *Cx)
{
    if (!Ap || !Ai || !Cp || !Ci) return 1;
    int *w = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    if (!w) return 1;
    for (int p = 0; p < Ap[n]; p++) w[Ai[p]]++;
    co_cumsum(Cp, w, m);
    for (int j = 0; j < n; j++)
        for (int p = Ap[j]; p < Ap[j + 1]; p++) {
            int q = w[Ai[p]]++;
            Ci[q] = j;
            if (Cx && Ax) Cx[q] = Ax[p];
        }
    free(w);
    return 0;
}

/* ---- csparse.py:1608-1642 + 1961-1989 ----------------------------------
 * C = A*B, A is m-by-k, B is k-by-n.  Cp (n+1 slots) is caller-owned; *Ci_out
 * and *Cx_out are malloc'ed here at exactly nnz(C) entries (the reference trims
 * C with cs_sprealloc(C, 0)) and released with co_free.  Ax or Bx NULL gives a
 * pattern-only product (*Cx_out = NULL). */
int co_multiply(int m, int k, int n, const int *Ap, const int *Ai, const double *Ax,
                const int *Bp, const int *Bi, const double *Bx,
                int *Cp, int **Ci_out, double **Cx_out)
{
    (void)k;
    if (!Ap || !Ai || !Bp || !Bi || !Cp || !Ci_out || !Cx_out) return 1;
    int values = (Ax != NULL) && (Bx != NULL);
    size_t cap = (size_t)Ap[k] + (size_t)Bp[n];
    if (cap < 1) cap = 1;
    int *w = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    double *x = values ? (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double)) : NULL;
    int *Ci = (int *)malloc(cap * sizeof(int));
    double *Cx = values ? (double *)malloc(cap * sizeof(double)) : NULL;
    size_t nz = 0;
    for (int j = 0; j < n; j++) {
        if (nz + (size_t)m > cap) {               /* csparse.py:1630-1631 */
            cap = 2 * cap + (size_t)m;
            Ci = (int *)realloc(Ci, cap * sizeof(int));
            if (values) Cx = (double *)realloc(Cx, cap * sizeof(double));
        }
        Cp[j] = (int)nz;
        int mark = j + 1;
        for (int pb = Bp[j]; pb < Bp[j + 1]; pb++) {
            int col = Bi[pb];
            double beta = Bx ? Bx[pb] : 1.0;
            for (int p = Ap[col]; p < Ap[col + 1]; p++) {
                int r = Ai[p];
                if (w[r] < mark) {
                    w[r] = mark;
                    Ci[nz++] = r;
                    if (values) x[r] = beta * Ax[p];
                } else if (values) {
                    double t = beta * Ax[p];
                    x[r] = x[r] + t;
                }
            }
        }
        if (values)
            for (size_t p = (size_t)Cp[j]; p < nz; p++) Cx[p] = x[Ci[p]];
    }
    Cp[n] = (int)nz;
    free(w);
    free(x);
    *Ci_out = Ci;
    *Cx_out = Cx;
    return 0;
}

void co_free(void *p) { free(p); }

/* ---- csparse.py:1330-1345 -------------------------------------------- */
int co_lsolve(int n, const int *Lp, const int *Li, const double *Lx, double *x)
{
    if (!Lp || !Li || !Lx || !x) return 1;
    for (int j = 0; j < n; j++) {
        if (Lx[Lp[j]] == 0.0) return 2;
        x[j] = x[j] / Lx[Lp[j]];
        double xj = x[j];
        for (int p = Lp[j] + 1; p < Lp[j + 1]; p++) {
            double t = Lx[p] * xj;
            x[Li[p]] = x[Li[p]] - t;
        }
    }
    return 0;
}

/* ---- csparse.py:1348-1365 -------------------------------------------- */
int co_ltsolve(int n, const int *Lp, const int *Li, const double *Lx, double *x)
{
    if (!Lp || !Li || !Lx || !x) return 1;
    for (int j = n - 1; j >= 0; j--) {
        for (int p = Lp[j] + 1; p < Lp[j + 1]; p++) {
            double t = Lx[p] * x[Li[p]];
            x[j] = x[j] - t;
        }
        if (Lx[Lp[j]] == 0.0) return 2;
        x[j] = x[j] / Lx[Lp[j]];
    }
    return 0;
}

/* ---- csparse.py:2368-2385 -------------------------------------------- */
int co_usolve(int n, const int *Up, const int *Ui, const double *Ux, double *x)
{
    if (!Up || !Ui || !Ux || !x) return 1;
    for (int j = n - 1; j >= 0; j--) {
        if (Ux[Up[j + 1] - 1] == 0.0) return 2;
        x[j] = x[j] / Ux[Up[j + 1] - 1];
        double xj = x[j];
        for (int p = Up[j]; p < Up[j + 1] - 1; p++) {
            double t = Ux[p] * xj;
            x[Ui[p]] = x[Ui[p]] - t;
        }
    }
    return 0;
}

/* ---- csparse.py:2460-2475 -------------------------------------------- */
int co_utsolve(int n, const int *Up, const int *Ui, const double *Ux, double *x)
{
    if (!Up || !Ui || !Ux || !x) return 1;
    for (int j = 0; j < n; j++) {
        for (int p = Up[j]; p < Up[j + 1] - 1; p++) {
            double t = Ux[p] * x[Ui[p]];
            x[j] = x[j] - t;
        }
        if (Ux[Up[j + 1] - 1] == 0.0) return 2;
        x[j] = x[j] / Ux[Up[j + 1] - 1];
    }
    return 0;
}

/* ---- csparse.py:1264-1277 / 1779-1792 -------------------------------- */
int co_ipvec(const int *p, const double *b, double *x, int n)
{
    if (!b || !x) return 1;
    for (int k = 0; k < n; k++) x[p ? p[k] : k] = b[k];
    return 0;
}

int co_pvec(const int *p, const double *b, double *x, int n)
{
    if (!b || !x) return 1;
    for (int k = 0; k < n; k++) x[k] = b[p ? p[k] : k];
    return 0;
}

/* ---- symbolic Cholesky, natural order -----------------------------------
 * csparse.py:2051-2072 (cs_schol) = cs_symperm(pattern of triu) :2220-2255,
 * cs_etree :1136-1169, cs_post/cs_tdfs :1711-1742/:2258-2289, cs_counts
 * :703-764 (LL'=A branch) with cs_leaf :1280-1304.  Input: any CSC matrix of
 * which only entries with row <= col are used.  Outputs: parent[n], cp[n+1]. */
static void etree_upper(int n, const int *Cp, const int *Ci, int *parent)
{
    int *anc = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int k = 0; k < n; k++) {
        parent[k] = -1;
        anc[k] = -1;
        for (int p = Cp[k]; p < Cp[k + 1]; p++) {
            int i = Ci[p];
            while (i != -1 && i < k) {
                int nxt = anc[i];
                anc[i] = k;
                if (nxt == -1) parent[i] = k;
                i = nxt;
            }
        }
    }
    free(anc);
}

static void postorder(int n, const int *parent, int *post)
{
    int *head = (int *)malloc(3 * (size_t)(n > 0 ? n : 1) * sizeof(int));
    int *nxt = head + n, *stack = head + 2 * n;
    for (int j = 0; j < n; j++) head[j] = -1;
    for (int j = n - 1; j >= 0; j--) {
        if (parent[j] == -1) continue;
        nxt[j] = head[parent[j]];
        head[parent[j]] = j;
    }
    int k = 0;
    for (int root = 0; root < n; root++) {
        if (parent[root] != -1) continue;
        int top = 0;
        stack[0] = root;
        while (top >= 0) {
            int node = stack[top], child = head[node];
            if (child == -1) {
                top--;
                post[k++] = node;
            } else {
                head[node] = nxt[child];
                stack[++top] = child;
            }
        }
    }
    free(head);
}

int co_schol(int n, const int *Ap, const int *Ai, int *parent, int *cp)
{
    if (!Ap || !Ai || !parent || !cp) return 1;
    size_t nn = (size_t)(n > 0 ? n : 1);
    /* upper-triangular pattern, columns in the order cs_symperm produces */
    int *Up = (int *)calloc(nn + 1, sizeof(int));
    int *w = (int *)calloc(nn, sizeof(int));
    for (int j = 0; j < n; j++)
        for (int p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] <= j) w[j]++;
    co_cumsum(Up, w, n);
    int *Ui = (int *)malloc(((size_t)Up[n] + 1) * sizeof(int));
    for (int j = 0; j < n; j++)
        for (int p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] <= j) Ui[w[j]++] = Ai[p];
    etree_upper(n, Up, Ui, parent);
    int *post = (int *)malloc(nn * sizeof(int));
    postorder(n, parent, post);
    /* transpose of the pattern: ATp/ATi */
    int *Tp = (int *)calloc(nn + 1, sizeof(int));
    int *Ti = (int *)malloc(((size_t)Up[n] + 1) * sizeof(int));
    co_transpose(n, n, Up, Ui, NULL, Tp, Ti, NULL);
    int *delta = (int *)calloc(nn, sizeof(int));
    int *first = (int *)malloc(4 * nn * sizeof(int));
    int *maxfirst = first + n, *prevleaf = first + 2 * n, *anc = first + 3 * n;
    for (int k = 0; k < n; k++) first[k] = maxfirst[k] = prevleaf[k] = -1;
    for (int k = 0; k < n; k++) {
        int j = post[k];
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    for (int i = 0; i < n; i++) anc[i] = i;
    for (int k = 0; k < n; k++) {
        int j = post[k];
        if (parent[j] != -1) delta[parent[j]]--;
        for (int p = Tp[j]; p < Tp[j + 1]; p++) {
            int i = Ti[p];
            if (i <= j || first[j] <= maxfirst[i]) continue;
            maxfirst[i] = first[j];
            int jprev = prevleaf[i];
            prevleaf[i] = j;
            delta[j]++;
            if (jprev != -1) {
                int q = jprev;
                while (q != anc[q]) q = anc[q];
                for (int s = jprev; s != q;) {
                    int sp = anc[s];
                    anc[s] = q;
                    s = sp;
                }
                delta[q]--;
            }
        }
        if (parent[j] != -1) anc[j] = parent[j];
    }
    for (int j = 0; j < n; j++)
        if (parent[j] != -1) delta[parent[j]] += delta[j];
    co_cumsum(cp, delta, n);
    free(Up); free(w); free(Ui); free(post); free(Tp); free(Ti); free(delta); free(first);
    return 0;
}

/* ---- csparse.py:561-619 with :1094-1131 inlined -------------------------
 * Up-looking numeric Cholesky, natural order.  Li/Lx hold cp[n] entries. */
int co_chol(int n, const int *Cp, const int *Ci, const double *Cx,
            const int *parent, const int *cp, int *Lp, int *Li, double *Lx)
{
    if (!Cp || !Ci || !Cx || !parent || !cp || !Lp || !Li || !Lx) return 1;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int *c = (int *)malloc(2 * nn * sizeof(int));
    int *s = c + n;
    char *mark = (char *)calloc(nn, 1);
    double *x = (double *)calloc(nn, sizeof(double));
    int status = 0;
    for (int k = 0; k < n; k++) Lp[k] = c[k] = cp[k];
    for (int k = 0; k < n && !status; k++) {
        int top = n;
        mark[k] = 1;
        for (int p = Cp[k]; p < Cp[k + 1]; p++) {
            int i = Ci[p], len = 0;
            if (i > k) continue;
            for (; !mark[i]; i = parent[i]) {
                s[len++] = i;
                mark[i] = 1;
            }
            while (len > 0) s[--top] = s[--len];
        }
        for (int p = top; p < n; p++) mark[s[p]] = 0;
        mark[k] = 0;
        x[k] = 0;
        for (int p = Cp[k]; p < Cp[k + 1]; p++)
            if (Ci[p] <= k) x[Ci[p]] = Cx[p];
        double d = x[k];
        x[k] = 0;
        for (; top < n; top++) {
            int i = s[top];
            double lki = x[i] / Lx[Lp[i]];
            x[i] = 0;
            for (int p = Lp[i] + 1; p < c[i]; p++) {
                double t = Lx[p] * lki;
                x[Li[p]] = x[Li[p]] - t;
            }
            double t2 = lki * lki;
            d = d - t2;
            int p = c[i]++;
            Li[p] = k;
            Lx[p] = lki;
        }
        if (d <= 0) { status = 3; break; }
        int p = c[k]++;
        Li[p] = k;
        Lx[p] = sqrt(d);
    }
    Lp[n] = cp[n];
    free(c); free(mark); free(x);
    return status;
}
