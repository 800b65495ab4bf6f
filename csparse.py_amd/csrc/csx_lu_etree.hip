// cs_lu (csparse.py:1370-1451, with cs_spsolve :2078-2113, cs_reach :1939-1958, cs_dfs :789-829) of ONE connected matrix
// on the device, natural column order, columns scheduled by the COLUMN ELIMINATION TREE (the elimination tree of A'A,
// csparse.py:1136-1169 with ata = True).
//
// Left-looking LU takes the columns one after the other: reach of A(:,k) in the graph of L, sparse triangular solve,
// threshold pivot search.  What column k really needs are the columns of L it reaches, and for any pivot sequence those
// are descendants of k in the column elimination tree (the structures of L and U are contained in the Cholesky factor
// of A'A).  Two columns that are not ancestor and descendant share no row (columns with a common row form a clique of
// A'A, hence lie on one root path), so they reach disjoint rows, pick their pivots among disjoint rows and append to
// L and U independently: every topological order of the tree gives the factors of the natural order, bit for bit.
//
// So: levels = heights in the column elimination tree; one launch per level, ONE LANE PER COLUMN running the host
// code's loop statement for statement (same DFS order, same order of the updates, true divisions, multiply and subtract
// rounded separately, first-maximum pivot with the diagonal preferred within tol), its work arrays (x, marks, DFS stack:
// 21 bytes per row) in a private slice of device memory that it leaves zeroed.  A column's entries go to a scratch strip
// reserved with one atomic per factor (the strip order depends on timing, the factors do not: they are assembled in
// column order afterwards, L's rows renumbered by pinv as the reference does last, :1447-1448).
// L (unit diagonal first), U (diagonal last) and pinv are bit-identical to csx_lu_host (tests/test_gpu_lu_etree.py).
//
// The gain is the tree's width: a chain (a banded matrix in natural order) has none, and the call hands back to the
// host code (*done = 0) when the tree is too deep for its size.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "csx_internal.h"

namespace csx {

namespace {

struct LuArgs {
    int32_t n;
    const int32_t *Ap, *Ai;
    const double *Ax;
    double tol;
    int32_t *pinv;                      // row -> pivot position, -1
    int32_t *Lstart, *Lend, *Ustart, *Uend;
    int32_t *Li, *Ui;                   // strips
    double *Lx, *Ux;
    unsigned long long *next;           // [0] next free entry of L's strip, [1] of U's
    unsigned long long lcap, ucap;
    int *flags;                         // [0] singular (min column), [1] strip overflow
    double *wx;                         // workspace: slot * n doubles
    int32_t *wi;                        // slot * 3 n ints: reach, stack, pos
    unsigned char *ws;                  // slot * n marks
};

#pragma clang fp contract(off)   // the host code rounds multiply and subtract separately (x86-64 baseline, no FMA)
__global__ __launch_bounds__(64) void k_lu_level(LuArgs a, const int32_t *__restrict__ cols, int32_t first, int32_t count) {
    const int32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= count) return;
    const int32_t n = a.n, k = cols[first + slot];
    double *x = a.wx + (size_t)slot * n;
    int32_t *reach = a.wi + (size_t)slot * 3 * n, *stack = reach + n, *pos = stack + n;
    unsigned char *seen = a.ws + (size_t)slot * n;
    // reach of A(:,k) in the graph of L: depth-first search, topological order in reach[top..n-1]
    int32_t top = n;
    for (int32_t p = a.Ap[k]; p < a.Ap[k + 1]; p++) {
        const int32_t r0 = a.Ai[p];
        if (seen[r0]) continue;
        int32_t head = 0;
        stack[0] = r0;
        while (head >= 0) {
            const int32_t jj = stack[head];
            const int32_t col = a.pinv[jj];
            if (!seen[jj]) {
                seen[jj] = 1;
                pos[head] = col < 0 ? 0 : a.Lstart[col];
            }
            bool done = true;
            const int32_t end = col < 0 ? 0 : a.Lend[col];
            for (int32_t q = pos[head]; q < end; q++) {
                const int32_t i = a.Li[q];
                if (seen[i]) continue;
                pos[head] = q;
                stack[++head] = i;
                done = false;
                break;
            }
            if (done) {
                head--;
                reach[--top] = jj;
            }
        }
    }
    int32_t nl = 0, nu = 1;
    for (int32_t p = top; p < n; p++) {
        const int32_t i = reach[p];
        seen[i] = 0;
        x[i] = 0.0;
        if (a.pinv[i] < 0) nl++;
        else nu++;
    }
    // room in the strips (the order of the reservations depends on timing; the assembled factors do not)
    const unsigned long long lo = atomicAdd(&a.next[0], (unsigned long long)nl), uo = atomicAdd(&a.next[1], (unsigned long long)nu);
    if (lo + (unsigned long long)nl > a.lcap || uo + (unsigned long long)nu > a.ucap) {
        atomicOr(&a.flags[1], 1);
        return;
    }
    for (int32_t p = a.Ap[k]; p < a.Ap[k + 1]; p++) x[a.Ai[p]] = a.Ax[p];
    // sparse triangular solve x = L \ A(:,k) along the reach
    for (int32_t px = top; px < n; px++) {
        const int32_t jj = reach[px], col = a.pinv[jj];
        if (col < 0) continue;
        const int32_t b = a.Lstart[col], e = a.Lend[col];
        x[jj] = x[jj] / a.Lx[b];
        const double xj = x[jj];
        for (int32_t q = b + 1; q < e; q++) {
            const double t = a.Lx[q] * xj;
            x[a.Li[q]] = x[a.Li[q]] - t;
        }
    }
    // pivot search among the non-pivotal rows (first maximum in reach order); pivotal rows go to U
    int32_t ipiv = -1, unz = (int32_t)uo, lnz = (int32_t)lo;
    double amax = -1.0;
    for (int32_t p = top; p < n; p++) {
        const int32_t i = reach[p];
        if (a.pinv[i] < 0) {
            const double t = fabs(x[i]);
            if (t > amax) {
                amax = t;
                ipiv = i;
            }
        } else {
            a.Ui[unz] = a.pinv[i];
            a.Ux[unz++] = x[i];
        }
    }
    if (ipiv == -1 || amax <= 0) {
        atomicMin(&a.flags[0], k);
        for (int32_t p = top; p < n; p++) x[reach[p]] = 0.0;
        return;
    }
    // the diagonal is preferred within tol (csparse.py:1426).  x[k] is non-zero only when row k is in this column's reach, and
    // then no other lane of the level owns it; a zero fails the test whatever pinv[k] says (tol > 0)
    if (fabs(x[k]) >= amax * a.tol && a.pinv[k] < 0) ipiv = k;
    const double pivot = x[ipiv];
    a.Ui[unz] = k;
    a.Ux[unz++] = pivot;
    a.Li[lnz] = ipiv;
    a.Lx[lnz++] = 1.0;
    for (int32_t p = top; p < n; p++) {
        const int32_t i = reach[p];
        if (a.pinv[i] < 0 && i != ipiv) {
            a.Li[lnz] = i;
            a.Lx[lnz++] = x[i] / pivot;
        }
        x[i] = 0.0;
    }
    a.Lstart[k] = (int32_t)lo;
    a.Lend[k] = lnz;
    a.Ustart[k] = (int32_t)uo;
    a.Uend[k] = unz;
    __threadfence();
    a.pinv[ipiv] = k;                   // rows of this column's subtree only: no other lane of the level reads it
}
#pragma clang fp contract(fast)

__global__ void k_lu_counts(int32_t n, const int32_t *__restrict__ s0, const int32_t *__restrict__ s1, int32_t *cnt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) cnt[j] = s1[j] - s0[j];
}

// one wave per column: strip -> its place in the factor; L's rows renumbered by pinv (csparse.py:1447-1448)
__global__ __launch_bounds__(256) void k_lu_assemble(int32_t n, const int32_t *__restrict__ start, const int32_t *__restrict__ Sp,
                                                     const int32_t *__restrict__ Si, const double *__restrict__ Sx,
                                                     const int32_t *__restrict__ relabel, const int32_t *__restrict__ Fp,
                                                     int32_t *Fi, double *Fx) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t b = start[j], cnt = Fp[j + 1] - Fp[j], d = Fp[j];
    (void)Sp;
    for (int32_t q = lane; q < cnt; q += 64) {
        const int32_t i = Si[b + q];
        Fi[d + q] = relabel ? relabel[i] : i;
        Fx[d + q] = Sx[b + q];
    }
}

}  // namespace

}  // namespace csx

using namespace csx;

extern "C" int csx_lu_etree(csx_handle_t hA, double tol, csx_handle_t *hL, csx_handle_t *hU, int32_t *pinv_host, int *done) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->x || A->m != A->n || !hL || !hU || !pinv_host || !done) return CSX_EINVAL;
    *done = 0;
    const int32_t n = A->n;
    if ((n < 2048 && ctx().opt.lu_etree != 2) || !(tol > 0.0) || !ctx().opt.lu_etree) return CSX_OK;
    CSX_TRY(csc_validate(A));
    hipStream_t s = ctx().stream;
    const bool say = std::getenv("CSX_CHOL_TIMING") != nullptr;
    // column elimination tree on the host (csparse.py:1136-1169, ata = True) from a copy of the pattern
    std::vector<int32_t> Ap((size_t)n + 1), Ai((size_t)std::max(A->nnz, 1));
    CSX_HIP(hipMemcpyAsync(Ap.data(), A->p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (A->nnz) CSX_HIP(hipMemcpyAsync(Ai.data(), A->i, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> parent((size_t)n, -1), anc((size_t)n, -1), prev((size_t)n, -1), height((size_t)n, 0);
    for (int32_t k = 0; k < n; k++)
        for (int32_t p = Ap[(size_t)k]; p < Ap[(size_t)k + 1]; p++) {
            int32_t i = prev[(size_t)Ai[(size_t)p]];
            while (i != -1 && i < k) {
                const int32_t up = anc[(size_t)i];
                anc[(size_t)i] = k;
                if (up == -1) parent[(size_t)i] = k;
                i = up;
            }
            prev[(size_t)Ai[(size_t)p]] = k;
        }
    int32_t nlev = 0;
    for (int32_t j = 0; j < n; j++) {
        if (parent[(size_t)j] >= 0) height[(size_t)parent[(size_t)j]] = std::max(height[(size_t)parent[(size_t)j]], height[(size_t)j] + 1);
        nlev = std::max(nlev, height[(size_t)j] + 1);
    }
    // What a lane can afford.  A lane walks its column's reach and the reached columns of L one dependent memory access at a
    // time (~1 us each on this chip against a few ns on a host core), and the levels run one after the other: the device
    // only wins when every column is cheap (a short reach: bounded by the column counts of the Cholesky factor of A'A,
    // csx_sqr_host) and the tree shallow.  Measured (profiles/r03_lu_connected.txt): a 300 x 300 unsymmetric grid in the
    // order-2 ordering (395 levels, counts up to ~1 800) takes 173 s here against 0.6 s for the host loop; W-chain (one
    // level per column: 89 000 levels) 4.9 s against 0.02 s.  Such matrices stay with the host ("lu.etree" = 2 overrides:
    // tests, the W-chain report).
    int32_t max_count = 0;
    if (ctx().opt.lu_etree != 2) {
        std::vector<int32_t> par2((size_t)n), cp((size_t)n), pv((size_t)2 * n), lm((size_t)n);
        int32_t m2 = 0;
        int64_t vnz = 0, rnz = 0;
        if (csx_sqr_host(n, n, Ap.data(), Ai.data(), par2.data(), cp.data(), pv.data(), lm.data(), &m2, &vnz, &rnz) != CSX_OK) return CSX_OK;
        for (int32_t j = 0; j < n; j++) max_count = std::max(max_count, cp[(size_t)j]);
    }
    if (say) std::fprintf(stderr, "csx_lu_etree: n %d, column elimination tree of %d levels, column counts up to %d\n", n, nlev, max_count);
    if (ctx().opt.lu_etree != 2 && (nlev > 96 || max_count > 96)) return CSX_OK;
    std::vector<int32_t> lev_ptr((size_t)nlev + 1, 0), order((size_t)n);
    for (int32_t j = 0; j < n; j++) lev_ptr[(size_t)height[(size_t)j] + 1]++;
    int32_t widest = 0;
    for (int32_t l = 0; l < nlev; l++) {
        widest = std::max(widest, lev_ptr[(size_t)l + 1]);
        lev_ptr[(size_t)l + 1] += lev_ptr[(size_t)l];
    }
    {
        std::vector<int32_t> at(lev_ptr.begin(), lev_ptr.end() - 1);
        for (int32_t j = 0; j < n; j++) order[(size_t)at[(size_t)height[(size_t)j]]++] = j;
    }
    // work space: 21 n bytes per column in flight, within a quarter of what is free (at most 16 GB)
    size_t free_b = 0, total_b = 0, idle_b = 0;
    pool_stats(&idle_b, nullptr);
    size_t budget = (size_t)16 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, (free_b + idle_b) / 4);
    int64_t slots = (int64_t)std::max<size_t>(64, budget / ((size_t)n * 21));
    slots = std::min<int64_t>(slots, ((int64_t)widest + 63) / 64 * 64);
    slots = slots / 64 * 64;
    DevScope tmp;
    LuArgs a{};
    a.n = n;
    a.Ap = A->p;
    a.Ai = A->i;
    a.Ax = A->x;
    a.tol = tol;
    int32_t *d_order = nullptr, *lcount = nullptr, *ucount = nullptr;
    CSX_TRY(tmp.alloc(&d_order, (size_t)n));
    CSX_HIP(hipMemcpyAsync(d_order, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_TRY(tmp.alloc(&a.pinv, (size_t)n));
    CSX_TRY(tmp.alloc(&a.Lstart, (size_t)n));
    CSX_TRY(tmp.alloc(&a.Lend, (size_t)n));
    CSX_TRY(tmp.alloc(&a.Ustart, (size_t)n));
    CSX_TRY(tmp.alloc(&a.Uend, (size_t)n));
    CSX_TRY(tmp.alloc(&lcount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&ucount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&a.next, 2));
    CSX_TRY(tmp.alloc(&a.flags, 2));
    CSX_TRY(tmp.alloc(&a.wx, (size_t)slots * n));
    CSX_TRY(tmp.alloc(&a.wi, (size_t)slots * 3 * n));
    CSX_TRY(tmp.alloc(&a.ws, (size_t)slots * n));
    CSX_HIP(hipMemsetAsync(a.wx, 0, (size_t)slots * n * sizeof(double), s));
    CSX_HIP(hipMemsetAsync(a.ws, 0, (size_t)slots * n, s));
    unsigned long long cap = (unsigned long long)A->nnz * 16 + (unsigned long long)n * 4;
    for (int attempt = 0; attempt < 4; attempt++, cap *= 4) {
        cap = std::min<unsigned long long>(cap, 0x7FFFFFF0ull);
        a.lcap = a.ucap = cap;
        int st = dalloc(&a.Li, (size_t)cap);
        if (st == CSX_OK) st = dalloc(&a.Lx, (size_t)cap);
        if (st == CSX_OK) st = dalloc(&a.Ui, (size_t)cap);
        if (st == CSX_OK) st = dalloc(&a.Ux, (size_t)cap);
        int hflags[2] = {0x7fffffff, 0};
        if (st == CSX_OK &&
            (hipMemsetAsync(a.pinv, 0xFF, (size_t)n * sizeof(int32_t), s) != hipSuccess ||
             hipMemsetAsync(a.next, 0, 2 * sizeof(unsigned long long), s) != hipSuccess ||
             hipMemcpyAsync(a.flags, hflags, sizeof hflags, hipMemcpyHostToDevice, s) != hipSuccess))
            st = CSX_ERUNTIME;
        for (int32_t l = 0; l < nlev && st == CSX_OK; l++)
            for (int32_t f = lev_ptr[(size_t)l]; f < lev_ptr[(size_t)l + 1]; f += (int32_t)slots) {
                const int32_t cnt = (int32_t)std::min<int64_t>(slots, (int64_t)lev_ptr[(size_t)l + 1] - f);
                hipLaunchKernelGGL(k_lu_level, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, s, a, d_order, f, cnt);
            }
        if (st == CSX_OK &&
            (hipGetLastError() != hipSuccess || hipMemcpyAsync(hflags, a.flags, sizeof hflags, hipMemcpyDeviceToHost, s) != hipSuccess ||
             hipStreamSynchronize(s) != hipSuccess))
            st = CSX_ERUNTIME;
        bool again = false;
        if (st == CSX_OK && hflags[0] != 0x7fffffff) st = CSX_ENOTSPD;     // singular: the reference returns None (csparse.py:1423)
        else if (st == CSX_OK && hflags[1]) again = cap < 0x7FFFFFF0ull;   // the strips were too small
        if (st == CSX_OK && hflags[1] && !again) st = CSX_EINVAL;
        if (st == CSX_OK && !again) {
            // assemble in column order
            Csc *L = new Csc(), *U = new Csc();
            L->m = L->n = U->m = U->n = n;
            int64_t lnz = 0, unz = 0;
            st = dalloc(&L->p, (size_t)n + 1);
            if (st == CSX_OK) st = dalloc(&U->p, (size_t)n + 1);
            const unsigned nb = (unsigned)(((int64_t)n + 255) / 256);
            hipLaunchKernelGGL(k_lu_counts, dim3(nb), dim3(256), 0, s, n, a.Lstart, a.Lend, lcount);
            hipLaunchKernelGGL(k_lu_counts, dim3(nb), dim3(256), 0, s, n, a.Ustart, a.Uend, ucount);
            if (st == CSX_OK) st = scan_exclusive_i32(lcount, L->p, n, &lnz);
            if (st == CSX_OK) st = scan_exclusive_i32(ucount, U->p, n, &unz);
            if (st == CSX_OK) {
                L->nnz = (int32_t)lnz;
                U->nnz = (int32_t)unz;
                st = dalloc(&L->i, (size_t)lnz);
                if (st == CSX_OK) st = dalloc(&L->x, (size_t)lnz);
                if (st == CSX_OK) st = dalloc(&U->i, (size_t)unz);
                if (st == CSX_OK) st = dalloc(&U->x, (size_t)unz);
            }
            if (st == CSX_OK) {
                const unsigned nw = (unsigned)(((int64_t)n + 3) / 4);
                hipLaunchKernelGGL(k_lu_assemble, dim3(nw), dim3(256), 0, s, n, a.Lstart, (const int32_t *)nullptr, a.Li, a.Lx, a.pinv, L->p,
                                   L->i, L->x);
                hipLaunchKernelGGL(k_lu_assemble, dim3(nw), dim3(256), 0, s, n, a.Ustart, (const int32_t *)nullptr, a.Ui, a.Ux,
                                   (const int32_t *)nullptr, U->p, U->i, U->x);
                if (hipGetLastError() != hipSuccess ||
                    hipMemcpyAsync(pinv_host, a.pinv, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess)
                    st = CSX_ERUNTIME;
            }
            if (st == CSX_OK) {
                *hL = put(K_CSC, L);
                *hU = put(K_CSC, U);
                *done = 1;
            } else {
                free_csc(L);
                free_csc(U);
            }
        }
        dfree(a.Li);
        dfree(a.Lx);
        dfree(a.Ui);
        dfree(a.Ux);
        if (!again) return st;
    }
    return CSX_EINVAL;
}
