// Pattern of the Cholesky factor on the device, for cs_chol (csparse.py:560-627).
//
// The reference finds row k of L with cs_ereach (csparse.py:847-878): from every entry A(i,k), i < k,
// walk up the elimination tree until a node already visited for this row; the visited nodes are the
// columns of row k, and cs_chol appends k to each of those columns as it goes (rows ascending inside a
// column, the diagonal first).  The marks make that walk inherently sequential.
//
// Here the marks are replaced by a test that needs no shared state.  Number the tree in postorder
// (every subtree is an interval [first[v], post[v]]) and sort the start nodes of row k by postorder:
//      i_0 < i_1 < ... (in postorder).
// The walk from i_t may stop at the first ancestor whose subtree contains i_{t-1} -- everything above
// it was visited from i_{t-1} -- i.e. at the first v on the path with first[v] <= post[i_{t-1}]
// (Gilbert, Ng & Peyton's row-subtree / skeleton argument).  So every (row, start) pair walks its own
// disjoint piece of the row subtree, all pairs in parallel, total work nnz(L).
//
//   1. starts: entries of P A P' above the diagonal as (row k, post[i]); two stable radix sorts group
//      them by k with post ascending inside a row                                   (csx_sort.hip)
//   2. walk once to count, scan, walk again to emit (column v, row k) pairs, plus the diagonals
//   3. stable sort by v  -> L.i (rows ascending inside a column, diagonal first) and L.p, which must
//      equal the S.cp the caller passed (else CSX_EINVAL: S does not belong to A)
//   4. stable sort of L's entries by row -> the row view used by the numeric kernels: for row j the
//      columns k (ascending, diagonal last) and the position of L(j,k) in L.i / L.x
//
// The host contributes only the O(n) postorder of the tree it already holds (S.parent).
#include <memory>
#include <vector>

#include "csx_internal.h"
#include "csx_sweep.h"
#include "csx_cholclique.h"

namespace csx {

// first[v], post[v]: the postorder interval of v's subtree; postinv[post[v]] = v.
// parent[v] > v for every non-root (elimination tree); returns false otherwise.
static bool postorder_intervals(int32_t n, const int32_t *parent, std::vector<int32_t> &first, std::vector<int32_t> &post,
                                std::vector<int32_t> &postinv) {
    std::vector<int32_t> size((size_t)n, 1), nf((size_t)n, 0);
    for (int32_t v = 0; v < n; v++) {
        const int32_t p = parent[v];
        if (p == -1) continue;
        if (p <= v || p >= n) return false;
        size[(size_t)p] += size[(size_t)v];
    }
    first.assign((size_t)n, 0);
    post.assign((size_t)n, 0);
    postinv.assign((size_t)n, 0);
    int32_t run = 0;
    for (int32_t v = n - 1; v >= 0; v--) {   // parents before children
        const int32_t p = parent[v];
        if (p < 0) {
            first[(size_t)v] = run;
            run += size[(size_t)v];
        } else {
            first[(size_t)v] = nf[(size_t)p];
            nf[(size_t)p] += size[(size_t)v];
        }
        nf[(size_t)v] = first[(size_t)v];
        post[(size_t)v] = first[(size_t)v] + size[(size_t)v] - 1;
        postinv[(size_t)post[(size_t)v]] = v;
    }
    return true;
}

// ---- 1. starts ---------------------------------------------------------------------------------------
// one wave per column c of A: entries with pinv[i] < pinv[c]
template <bool FILL>
__global__ __launch_bounds__(256) void k_sym_starts(int32_t n, const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                    const int32_t *__restrict__ pinv, const int32_t *__restrict__ post,
                                                    int32_t *__restrict__ cnt, const int32_t *__restrict__ sptr,
                                                    uint32_t *__restrict__ skey, uint32_t *__restrict__ srow, int *bad) {
    const int lane = threadIdx.x & 63;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= n) return;
    const int32_t k = pinv ? pinv[c] : (int32_t)c;
    const int32_t b = Ap[c], e = Ap[c + 1];
    int32_t run = FILL ? sptr[c] : 0;
    for (int32_t p0 = b; p0 < e; p0 += 64) {
        const int32_t p = p0 + lane;
        bool up = false;
        int32_t i = 0;
        if (p < e) {
            i = Ai[p];
            if (i < 0 || i >= n) {
                *bad = 1;
            } else {
                if (pinv) i = pinv[i];
                up = i < k;
            }
        }
        const unsigned long long bal = __ballot(up);
        if (FILL && up) {
            const int32_t q = run + __popcll(bal & ((1ull << lane) - 1ull));
            skey[q] = (uint32_t)post[i];
            srow[q] = (uint32_t)k;
        }
        run += __popcll(bal);
    }
    if (!FILL && lane == 0) cnt[c] = run;
}

// ---- 2. walks ----------------------------------------------------------------------------------------
// item layout of the combined count / offset array: row k owns items [k + sptr[k], k + 1 + sptr[k+1]):
// its diagonal first, then its starts.
__global__ __launch_bounds__(256) void k_sym_diag_items(int32_t n, const int32_t *__restrict__ sptr, int32_t *__restrict__ items) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) items[k + sptr[k]] = 1;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_sym_walk(int64_t ns, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ posts,
                                                  const int32_t *__restrict__ postinv, const int32_t *__restrict__ first,
                                                  const int32_t *__restrict__ parent, int32_t *__restrict__ items,
                                                  uint32_t *__restrict__ ev, uint32_t *__restrict__ ek) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns) return;
    const int32_t k = (int32_t)rows[t];
    const bool has_prev = t > 0 && rows[t - 1] == (uint32_t)k;
    const int32_t prev_post = has_prev ? (int32_t)posts[t - 1] : -1;
    int32_t v = postinv[posts[t]];
    const int64_t item = t + k + 1;
    int64_t out = EMIT ? items[item] : 0;
    int32_t c = 0;
    // ancestors of a start of row k are < k until k itself is reached (parent[v] > v was checked on the host)
    while (v >= 0 && v < k && first[v] > prev_post) {
        if (EMIT) {
            ev[out + c] = (uint32_t)v;
            if (ek) ek[out + c] = (uint32_t)k;
        }
        c++;
        v = parent[v];
    }
    if (!EMIT) items[item] = c;
}

__global__ __launch_bounds__(256) void k_sym_emit_diag(int32_t n, const int32_t *__restrict__ sptr, const int32_t *__restrict__ items,
                                                       uint32_t *__restrict__ ev, uint32_t *__restrict__ ek) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t out = items[k + sptr[k]];
    ev[out] = (uint32_t)k;
    if (ek) ek[out] = (uint32_t)k;
}

// ---- 3./4. checks and the row view --------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sym_compare(int64_t n1, const int32_t *__restrict__ a, const int32_t *__restrict__ b, int *bad) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n1 && a[q] != b[q]) *bad = 1;
}

// payload of the second sort: (column of the entry, its position in L) in one 64-bit word
__global__ __launch_bounds__(256) void k_sym_pack(int64_t lnz, const uint32_t *__restrict__ col, double *__restrict__ packed) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < lnz) packed[p] = __longlong_as_double((long long)(((unsigned long long)p << 32) | (unsigned long long)col[p]));
}

__global__ __launch_bounds__(256) void k_sym_unpack(int64_t lnz, const double *__restrict__ packed, int32_t *__restrict__ row_col,
                                                    int32_t *__restrict__ row_pos) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= lnz) return;
    const unsigned long long w = (unsigned long long)__double_as_longlong(packed[q]);
    row_col[q] = (int32_t)(w & 0xFFFFFFFFull);
    row_pos[q] = (int32_t)(w >> 32);
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// On success the five outputs are device arrays owned by the caller (dfree):
//   Lp (n+1), Li (lnz), row_ptr (n+1), row_col (lnz), row_pos (lnz); the row view includes the diagonal
//   as the LAST entry of each row.
// Counts-only mode (cs_schol): cp == nullptr and cp_host_out != nullptr -- the column counts of L are
// computed from the same walks (emit the column of every entry, sort, boundaries) and returned on the
// host as column pointers; the five device outputs are not produced.
int chol_symbolic_device(const Csc *A, const int32_t *parent, const int32_t *cp, const int32_t *pinv, int32_t **Lp_out,
                         int32_t **Li_out, int32_t **row_ptr_out, int32_t **row_col_out, int32_t **row_pos_out,
                         int32_t *cp_host_out) {
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    const bool counts_only = cp == nullptr;
    if (counts_only && !cp_host_out) return CSX_EINVAL;
    int64_t lnz = counts_only ? -1 : cp[n];
    if (!counts_only) {
        if (lnz < n) return CSX_EINVAL;
        for (int32_t j = 0; j < n; j++)
            if (cp[j + 1] - cp[j] < 1) return CSX_EINVAL;
    }
    std::vector<int32_t> first, post, postinv;
    if (!postorder_intervals(n, parent, first, post, postinv)) return CSX_EINVAL;
    if (pinv) {   // must be a permutation of 0..n-1
        std::vector<char> seen((size_t)n, 0);
        for (int32_t j = 0; j < n; j++) {
            if (pinv[j] < 0 || pinv[j] >= n || seen[(size_t)pinv[j]]) return CSX_EINVAL;
            seen[(size_t)pinv[j]] = 1;
        }
    }
    int32_t *d_parent = nullptr, *d_first = nullptr, *d_post = nullptr, *d_postinv = nullptr, *d_pinv = nullptr, *d_cp = nullptr;
    int32_t *cnt = nullptr, *sptr0 = nullptr, *sptr = nullptr, *items = nullptr;
    uint32_t *skey = nullptr, *srow = nullptr, *k1 = nullptr, *r1 = nullptr, *rows = nullptr, *posts = nullptr;
    uint32_t *ev = nullptr, *ek = nullptr, *sv = nullptr, *sk2 = nullptr;
    int32_t *Lp = nullptr, *Li = nullptr, *row_ptr = nullptr, *row_col = nullptr, *row_pos = nullptr;
    double *packed = nullptr, *packed_s = nullptr;
    int *bad = nullptr;
    int hbad = 0;
    int64_t ns = 0, total = 0;
    auto up = [&](int32_t **d, const int32_t *h, size_t count) {
        int st = dalloc(d, count);
        if (st == CSX_OK && count &&
            hipMemcpyAsync(*d, h, count * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess)
            st = CSX_ERUNTIME;
        return st;
    };
    int st = up(&d_parent, parent, (size_t)n);
    if (st == CSX_OK) st = up(&d_first, first.data(), (size_t)n);
    if (st == CSX_OK) st = up(&d_post, post.data(), (size_t)n);
    if (st == CSX_OK) st = up(&d_postinv, postinv.data(), (size_t)n);
    if (st == CSX_OK && !counts_only) st = up(&d_cp, cp, (size_t)n + 1);
    if (st == CSX_OK && pinv) st = up(&d_pinv, pinv, (size_t)n);
    if (st == CSX_OK) st = dalloc(&bad, 1);
    if (st == CSX_OK) st = dalloc(&cnt, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&sptr0, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&sptr, (size_t)n + 1);
    if (st == CSX_OK) {
        (void)hipMemsetAsync(bad, 0, sizeof(int), s);
        hipLaunchKernelGGL(k_sym_starts<false>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, A->p, A->i, d_pinv,
                           d_post, cnt, nullptr, nullptr, nullptr, bad);
        st = scan_exclusive_i32(cnt, sptr0, n, &ns);
    }
    // ---- 1. starts grouped by row, postorder ascending inside a row ----
    if (st == CSX_OK) st = dalloc(&skey, (size_t)ns);
    if (st == CSX_OK) st = dalloc(&srow, (size_t)ns);
    if (st == CSX_OK) st = dalloc(&k1, (size_t)ns);
    if (st == CSX_OK) st = dalloc(&r1, (size_t)ns);
    if (st == CSX_OK) st = dalloc(&rows, (size_t)ns);
    if (st == CSX_OK) st = dalloc(&posts, (size_t)ns);
    if (st == CSX_OK && ns > 0) {
        hipLaunchKernelGGL(k_sym_starts<true>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, A->p, A->i, d_pinv,
                           d_post, nullptr, sptr0, skey, srow, bad);
        st = stable_sort_by_key(skey, srow, nullptr, ns, (uint32_t)n, k1, r1, nullptr);
        if (st == CSX_OK) st = stable_sort_by_key(r1, k1, nullptr, ns, (uint32_t)n, rows, posts, nullptr);
    }
    if (st == CSX_OK) {
        if (ns > 0) st = boundaries_from_sorted(rows, ns, n, sptr);
        else (void)hipMemsetAsync(sptr, 0, ((size_t)n + 1) * sizeof(int32_t), s);
    }
    // ---- 2. count, scan, emit ----
    const int64_t nitems = ns + n;
    if (st == CSX_OK) st = dalloc(&items, (size_t)nitems + 1);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_sym_diag_items, dim3(blocks_for(n)), dim3(256), 0, s, n, sptr, items);
        if (ns > 0)
            hipLaunchKernelGGL(k_sym_walk<false>, dim3(blocks_for(ns)), dim3(256), 0, s, ns, rows, posts, d_postinv, d_first,
                               d_parent, items, nullptr, nullptr);
        st = scan_exclusive_i32(items, items, nitems, &total);
    }
    if (st == CSX_OK && counts_only) {
        if (total > 0x7FFFFFFFll) st = CSX_EINVAL;   // L does not fit int32 indices
        lnz = total;
    }
    if (st == CSX_OK && total != lnz) st = CSX_EINVAL;   // S.cp / S.parent do not describe chol(A)
    if (st == CSX_OK) st = dalloc(&ev, (size_t)lnz);
    if (st == CSX_OK && !counts_only) st = dalloc(&ek, (size_t)lnz);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_sym_emit_diag, dim3(blocks_for(n)), dim3(256), 0, s, n, sptr, items, ev, ek);
        if (ns > 0)
            hipLaunchKernelGGL(k_sym_walk<true>, dim3(blocks_for(ns)), dim3(256), 0, s, ns, rows, posts, d_postinv, d_first,
                               d_parent, items, ev, ek);
    }
    if (st == CSX_OK && counts_only) {   // column counts: sort the columns alone, boundaries = column pointers
        st = dalloc(&sv, (size_t)lnz);
        if (st == CSX_OK) st = dalloc(&Lp, (size_t)n + 1);
        if (st == CSX_OK) st = stable_sort_by_key(ev, nullptr, nullptr, lnz, (uint32_t)n, sv, nullptr, nullptr);
        if (st == CSX_OK) st = boundaries_from_sorted(sv, lnz, n, Lp);
        if (st == CSX_OK &&
            (hipMemcpyAsync(cp_host_out, Lp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
             hipMemcpyAsync(&hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s) != hipSuccess ||
             hipStreamSynchronize(s) != hipSuccess))
            st = CSX_ERUNTIME;
        if (st == CSX_OK && hbad) st = CSX_EINVAL;
        for (void *p : {(void *)d_parent, (void *)d_first, (void *)d_post, (void *)d_postinv, (void *)d_pinv, (void *)cnt,
                        (void *)sptr0, (void *)sptr, (void *)items, (void *)skey, (void *)srow, (void *)k1, (void *)r1,
                        (void *)rows, (void *)posts, (void *)ev, (void *)sv, (void *)Lp, (void *)bad})
            dfree(p);
        return st;
    }
    // ---- 3. L in CSC ----
    if (st == CSX_OK) st = dalloc(&sv, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&Li, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&Lp, (size_t)n + 1);
    if (st == CSX_OK) st = stable_sort_by_key(ev, ek, nullptr, lnz, (uint32_t)n, sv, (uint32_t *)Li, nullptr);
    if (st == CSX_OK) st = boundaries_from_sorted(sv, lnz, n, Lp);
    if (st == CSX_OK)
        hipLaunchKernelGGL(k_sym_compare, dim3(blocks_for((int64_t)n + 1)), dim3(256), 0, s, (int64_t)n + 1, Lp, d_cp, bad);
    // ---- 4. row view ----
    if (st == CSX_OK) st = dalloc(&packed, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&packed_s, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&sk2, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&row_ptr, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&row_col, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&row_pos, (size_t)lnz);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_sym_pack, dim3(blocks_for(lnz)), dim3(256), 0, s, lnz, sv, packed);
        st = stable_sort_by_key((const uint32_t *)Li, nullptr, packed, lnz, (uint32_t)n, sk2, nullptr, packed_s);
    }
    if (st == CSX_OK) st = boundaries_from_sorted(sk2, lnz, n, row_ptr);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_sym_unpack, dim3(blocks_for(lnz)), dim3(256), 0, s, lnz, packed_s, row_col, row_pos);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            set_error("cs_chol (pattern): %s", hipGetErrorString(hipGetLastError()));
            st = CSX_ERUNTIME;
        }
    }
    if (st == CSX_OK && hbad) st = CSX_EINVAL;
    for (void *p : {(void *)d_parent, (void *)d_first, (void *)d_post, (void *)d_postinv, (void *)d_pinv, (void *)d_cp,
                    (void *)cnt, (void *)sptr0, (void *)sptr, (void *)items, (void *)skey, (void *)srow, (void *)k1,
                    (void *)r1, (void *)rows, (void *)posts, (void *)ev, (void *)ek, (void *)sv, (void *)sk2,
                    (void *)packed, (void *)packed_s, (void *)bad})
        dfree(p);
    if (st != CSX_OK) {
        dfree(Lp);
        dfree(Li);
        dfree(row_ptr);
        dfree(row_col);
        dfree(row_pos);
        return st;
    }
    *Lp_out = Lp;
    *Li_out = Li;
    *row_ptr_out = row_ptr;
    *row_col_out = row_col;
    *row_pos_out = row_pos;
    return CSX_OK;
}

// csx_host.cpp
void etree_of_csc(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent);

}  // namespace csx

using namespace csx;

// cs_schol with natural ordering (csparse.py:2051-2072) for a device-resident matrix: the elimination tree
// on the host (Liu's algorithm with path compression is sequential and O(nnz)), the column counts on the
// device from the same row-subtree walks cs_chol uses.  parent[n], cp[n+1] are host arrays.
// Elimination tree (csparse.py:1136-1169, ata = False) of a matrix whose graph falls into many small connected
// components: Liu's algorithm is sequential in the columns of ONE component but components do not interact,
// so each gets a thread that runs the reference's loop (ancestor path compression included) on its own
// columns in ascending order.  G-spd (78 125 blocks of 64): 0.6 s on one host core incl. the copy -> see DESIGN.md.
__global__ __launch_bounds__(64) void k_etree_components(int32_t ncomp, const Tree *__restrict__ comps,
                                                         const uint32_t *__restrict__ nodes,
                                                         const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                         int32_t *parent, int32_t *ancestor) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncomp) return;
    const Tree t = comps[c];
    for (int32_t a = 0; a < t.count; a++) {
        const int32_t k = (int32_t)nodes[t.first + a];
        parent[k] = -1;
        ancestor[k] = -1;
        for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) {
            int32_t i = Ai[p];
            while (i != -1 && i < k) {          // the rows of column k above the diagonal are in k's component
                const int32_t inext = ancestor[i];
                ancestor[i] = k;
                if (inext == -1) parent[i] = k;
                i = inext;
            }
        }
    }
}

constexpr int ETREE_COMP_MAX = 2048;    // columns per component for the one-thread-per-component kernel
constexpr int ETREE_COMP_MIN_COUNT = 2048;

// *done = true when the tree was built on the device (parent[] on the host filled)
static int etree_by_components(const Csc *A, int32_t *parent_host, bool *done, bool *bad_index) {
    *done = false;
    *bad_index = false;
    const int32_t n = A->n;
    if (n < ETREE_COMP_MIN_COUNT) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *root = nullptr, *d_parent = nullptr, *d_anc = nullptr;
    uint32_t *nodes = nullptr;
    CSX_TRY(tmp.alloc(&root, (size_t)n));
    CSX_TRY(connected_components(n, A->p, A->i, 0, 0, 0, root, bad_index));
    if (*bad_index) return CSX_OK;
    CSX_TRY(tmp.alloc(&nodes, (size_t)n));
    Tree *comps = nullptr;
    int32_t ncomp = 0, maxc = 0;
    int st = group_by_root(n, root, nodes, nullptr, &comps, &ncomp, &maxc);
    if (st == CSX_OK && ncomp >= ETREE_COMP_MIN_COUNT && maxc <= ETREE_COMP_MAX) {
        st = tmp.alloc(&d_parent, (size_t)n);
        if (st == CSX_OK) st = tmp.alloc(&d_anc, (size_t)n);
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_etree_components, dim3((unsigned)((ncomp + 63) / 64)), dim3(64), 0, s, ncomp, comps, nodes,
                               A->p, A->i, d_parent, d_anc);
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(parent_host, d_parent, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
            else
                *done = true;
        }
    }
    dfree(comps);
    return st;
}

extern "C" int csx_schol(csx_handle_t hA, int32_t *parent, int32_t *cp) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !parent || !cp || A->m != A->n) return CSX_EINVAL;
    const int32_t n = A->n;
    if (n == 0) {
        cp[0] = 0;
        return CSX_OK;
    }
    hipStream_t s = ctx().stream;
    if (ctx().opt.chol_clique) {
        // a forest of cliques on consecutive columns (block-diagonal with dense blocks): tree and counts follow from the
        // smallest upper row of every column, one pass over A's pattern (csx_cholclique.hip)
        CliqueForest F;
        bool ok = false;
        int st = clique_forest(A, &F, &ok);
        if (st == CSX_OK && ok) {
            if (hipMemcpyAsync(parent, F.parent, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(cp, F.cp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        if (st == CSX_OK && ok) {             // kept on the matrix for the csx_chol that follows (csx_csc_invalidate drops it)
            free_clique_cache(A->clique);
            A->clique = new CliqueForest(F);
            return CSX_OK;
        }
        free_clique(&F);
        if (st != CSX_OK) return st;
    }
    free_clique_cache(A->clique);             // no finding under the options in force: none of an earlier call either
    A->clique = nullptr;
    bool on_device = false, bad_index = false;
    CSX_TRY(etree_by_components(A, parent, &on_device, &bad_index));
    if (bad_index) return CSX_EINVAL;
    if (!on_device) {   // one big component (or a small matrix): Liu's algorithm on the host
        std::unique_ptr<int32_t[]> hp(new int32_t[(size_t)n + 1]), hi(new int32_t[(size_t)A->nnz + 1]);
        CSX_HIP(hipMemcpyAsync(hp.get(), A->p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (A->nnz) CSX_HIP(hipMemcpyAsync(hi.get(), A->i, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        for (int32_t q = 0; q < A->nnz; q++)
            if (hi[(size_t)q] < 0 || hi[(size_t)q] >= n) return CSX_EINVAL;
        etree_of_csc(n, hp.get(), hi.get(), parent);
    }
    return chol_symbolic_device(A, parent, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cp);
}
