// cs_spsolve (csparse.py:2078-2113), cs_reach (:1939-1958), cs_dfs (:789-829) for ALL columns of B at once:
// X(:,k) = G \ B(:,k) with G lower (lo) or upper triangular, B sparse, optionally through the row permutation pinv
// of an LU in progress (column J = pinv[j] of G belongs to node j; J < 0: no column yet).
//
// The reference solves one column: a depth-first search of G's graph from the rows of B(:,k) leaves the reach in
// xi[top..n-1] in topological order -- the order is part of the result, cs_lu walks it -- and the numeric loop visits
// exactly those columns.  Columns of B are independent, so the device runs the reference's own sequential loops, one
// lane per column of B: the same stack discipline (xi[0..head] grows up, the output xi[top..n-1] grows down in the
// same array, pstack = xi + n), a byte per node instead of the sign flip of G.p, the same order of operations in the
// numeric part (true division, multiply and subtract rounded separately).  X comes back as a CSC matrix whose column k
// lists xi[top..n-1] in that order with x[xi[p]] beside it: pattern and values bit-identical to the reference's.
//
// Work space per column in flight: 2n ints + n bytes (+ n doubles with values), per-lane contiguous (a lane's stack
// top stays in its own cache lines); the columns of B are taken in chunks sized to a memory budget.
#include <algorithm>
#include <vector>

#include "csx_internal.h"

namespace csx {

constexpr size_t SPS_BUDGET = (size_t)32 << 30;  // work space of the columns in flight: at most this, and a quarter of what is free
constexpr int64_t SPS_MAX_LANES = 1 << 16;

__global__ __launch_bounds__(64) void k_sps_reach(int32_t n, const int32_t *__restrict__ Gp,
                                                  const int32_t *__restrict__ Gi, const int32_t *__restrict__ pinv,
                                                  const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                  int32_t k0, int32_t nc, int32_t *xi_all, unsigned char *mark_all,
                                                  int32_t *top_out, int32_t *count_out) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= nc) return;
    const int32_t k = k0 + (int32_t)lane;
    int32_t *xi = xi_all + lane * 2 * (int64_t)n, *pstack = xi + n;
    unsigned char *mark = mark_all + lane * (int64_t)n;   // all zero on entry, all zero again on exit
    int32_t top = n;
    for (int32_t pb = Bp[k]; pb < Bp[k + 1]; pb++) {      // cs_reach :1950-1953
        const int32_t start = Bi[pb];
        if (mark[start]) continue;
        int32_t head = 0;                                 // cs_dfs :803-829
        xi[0] = start;
        while (head >= 0) {
            const int32_t j = xi[head];
            const int32_t jnew = pinv ? pinv[j] : j;
            if (!mark[j]) {
                mark[j] = 1;
                pstack[head] = jnew < 0 ? 0 : Gp[jnew];
            }
            bool done = true;
            const int32_t p2 = jnew < 0 ? 0 : Gp[jnew + 1];
            for (int32_t p = pstack[head]; p < p2; p++) {
                const int32_t i = Gi[p];
                if (mark[i]) continue;
                pstack[head] = p;
                xi[++head] = i;
                done = false;
                break;
            }
            if (done) {
                head--;
                xi[--top] = j;
            }
        }
    }
    for (int32_t p = top; p < n; p++) mark[xi[p]] = 0;    // :1955-1956 restores G.p
    top_out[lane] = top;
    count_out[lane] = n - top;
}

#pragma clang fp contract(off)
template <bool VALUES>
__global__ __launch_bounds__(64) void k_sps_numeric(int32_t n, const int32_t *__restrict__ Gp,
                                                    const int32_t *__restrict__ Gi, const double *__restrict__ Gx,
                                                    const int32_t *__restrict__ pinv, int lo,
                                                    const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                    const double *__restrict__ Bx, int32_t k0, int32_t nc,
                                                    const int32_t *xi_all, double *x_all, const int32_t *top_in,
                                                    const int32_t *__restrict__ Xp, int32_t *Xi, double *Xx) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= nc) return;
    const int32_t k = k0 + (int32_t)lane;
    const int32_t *xi = xi_all + lane * 2 * (int64_t)n;
    const int32_t top = top_in[lane];
    const int64_t out = Xp[lane];
    if (VALUES) {
        double *x = x_all + lane * (int64_t)n;
        for (int32_t p = top; p < n; p++) x[xi[p]] = 0.0;                        // :2093-2094
        for (int32_t p = Bp[k]; p < Bp[k + 1]; p++) x[Bi[p]] = Bx[p];            // :2095-2096
        for (int32_t px = top; px < n; px++) {                                   // :2097-2112
            const int32_t j = xi[px];
            const int32_t J = pinv ? pinv[j] : j;
            if (J < 0) continue;
            const double xj = x[j] / Gx[lo ? Gp[J] : Gp[J + 1] - 1];
            x[j] = xj;
            const int32_t p = lo ? Gp[J] + 1 : Gp[J], q = lo ? Gp[J + 1] : Gp[J + 1] - 1;
            for (int32_t t = p; t < q; t++) {
                const double prod = Gx[t] * xj;
                x[Gi[t]] = x[Gi[t]] - prod;
            }
        }
        for (int32_t p = top; p < n; p++) {
            Xi[out + (p - top)] = xi[p];
            Xx[out + (p - top)] = x[xi[p]];
        }
    } else {
        for (int32_t p = top; p < n; p++) Xi[out + (p - top)] = xi[p];
    }
}
#pragma clang fp contract(fast)

__global__ void k_sps_shift(int32_t *p, int64_t n, int32_t by) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += by;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_spsolve(csx_handle_t hG, csx_handle_t hB, const int32_t *pinv_host, int lo, int values,
                           csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *G = csc(hG), *B = csc(hB);
    if (!G || !B || !out || G->m != G->n || B->m != G->n) return CSX_EINVAL;
    const bool with_values = values != 0;
    if (with_values && (!G->x || !B->x)) return CSX_EINVAL;
    // the reach kernels index per-lane work arrays with G's and B's row indices: arrays the library did not make are checked
    CSX_TRY(csc_validate(G));
    CSX_TRY(csc_validate(B));
    const int32_t n = G->n, nb = B->n;
    if (pinv_host)
        for (int32_t j = 0; j < n; j++)
            if (pinv_host[j] >= n) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    Csc *X = new Csc();
    X->m = n;
    X->n = nb;
    X->owns = true;
    int st = dalloc(&X->p, (size_t)nb + 1);
    struct Piece {
        int32_t *i;
        double *x;
        int32_t k0, nc;
        int64_t nnz;
    };
    std::vector<Piece> pieces;
    DevScope tmp;
    int32_t *d_pinv = nullptr, *xi = nullptr, *top = nullptr, *cnt = nullptr;
    unsigned char *mark = nullptr;
    double *xw = nullptr;
    int64_t total = 0;
    if (st == CSX_OK && (n == 0 || nb == 0)) {
        CSX_HIP(hipMemsetAsync(X->p, 0, ((size_t)nb + 1) * sizeof(int32_t), s));
    } else if (st == CSX_OK) {
        if (pinv_host) {
            st = tmp.alloc(&d_pinv, (size_t)n);
            if (st == CSX_OK && hipMemcpyAsync(d_pinv, pinv_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        const size_t per_lane = (size_t)n * (8 + 1 + (with_values ? 8 : 0));
        size_t budget = SPS_BUDGET, free_b = 0, total_b = 0, idle_b = 0;
        pool_stats(&idle_b, nullptr);   // idle blocks of the caching allocator are reusable
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, (free_b + idle_b) / 4);
        int64_t lanes = (int64_t)std::max<size_t>(64, budget / std::max<size_t>(per_lane, 1));
        lanes = std::min<int64_t>(std::min<int64_t>(lanes, SPS_MAX_LANES), ((int64_t)nb + 63) / 64 * 64);
        if (st == CSX_OK) st = tmp.alloc(&xi, (size_t)lanes * 2 * n);
        if (st == CSX_OK) st = tmp.alloc(&mark, (size_t)lanes * n);
        if (st == CSX_OK && with_values) st = tmp.alloc(&xw, (size_t)lanes * n);
        if (st == CSX_OK) st = tmp.alloc(&top, (size_t)lanes);
        if (st == CSX_OK) st = tmp.alloc(&cnt, (size_t)lanes + 1);
        if (st == CSX_OK && hipMemsetAsync(mark, 0, (size_t)lanes * n, s) != hipSuccess) st = CSX_ERUNTIME;
        for (int32_t k0 = 0; k0 < nb && st == CSX_OK; k0 += (int32_t)lanes) {
            const int32_t nc = (int32_t)std::min<int64_t>(lanes, (int64_t)nb - k0);
            const unsigned grid = (unsigned)((nc + 63) / 64);
            hipLaunchKernelGGL(k_sps_reach, dim3(grid), dim3(64), 0, s, n, G->p, G->i, d_pinv, B->p, B->i, k0, nc, xi, mark, top,
                               cnt);
            if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
            int64_t piece_nnz = 0;
            // this chunk's column pointers straight into X.p (relative to the chunk; shifted below)
            if (st == CSX_OK) st = scan_exclusive_i32(cnt, X->p + k0, nc, &piece_nnz);
            if (st != CSX_OK) break;
            if (total + piece_nnz > 0x7fffffff) {
                set_error("csx_spsolve: more than 2^31 - 1 entries in X");
                st = CSX_ERUNTIME;
                break;
            }
            Piece pc{nullptr, nullptr, k0, nc, piece_nnz};
            st = dalloc(&pc.i, (size_t)piece_nnz);
            if (st == CSX_OK && with_values) st = dalloc(&pc.x, (size_t)piece_nnz);
            pieces.push_back(pc);
            if (st != CSX_OK) break;
            if (with_values)
                hipLaunchKernelGGL(k_sps_numeric<true>, dim3(grid), dim3(64), 0, s, n, G->p, G->i, G->x, d_pinv, lo, B->p, B->i,
                                   B->x, k0, nc, xi, xw, top, X->p + k0, pc.i, pc.x);
            else
                hipLaunchKernelGGL(k_sps_numeric<false>, dim3(grid), dim3(64), 0, s, n, G->p, G->i, G->x, d_pinv, lo, B->p, B->i,
                                   B->x, k0, nc, xi, xw, top, X->p + k0, pc.i, pc.x);
            if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
            if (st == CSX_OK && total > 0)
                hipLaunchKernelGGL(k_sps_shift, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, X->p + k0, (int64_t)nc,
                                   (int32_t)total);
            total += piece_nnz;
        }
        if (st == CSX_OK) {
            const int32_t t32 = (int32_t)total;
            if (hipMemcpyAsync(X->p + nb, &t32, sizeof t32, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
    }
    // one chunk: its arrays are X's; several: concatenate
    if (st == CSX_OK) {
        X->nnz = (int32_t)total;
        if (pieces.size() == 1) {
            X->i = pieces[0].i;
            X->x = pieces[0].x;
            pieces.clear();
        } else {
            st = dalloc(&X->i, (size_t)total);
            if (st == CSX_OK && with_values) st = dalloc(&X->x, (size_t)total);
            int64_t off = 0;
            for (const Piece &pc : pieces) {
                if (st != CSX_OK) break;
                if (pc.nnz && hipMemcpyAsync(X->i + off, pc.i, (size_t)pc.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
                    st = CSX_ERUNTIME;
                if (pc.nnz && with_values &&
                    hipMemcpyAsync(X->x + off, pc.x, (size_t)pc.nnz * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
                    st = CSX_ERUNTIME;
                off += pc.nnz;
            }
        }
    }
    if (hipStreamSynchronize(s) != hipSuccess && st == CSX_OK) st = CSX_ERUNTIME;
    for (const Piece &pc : pieces) {
        dfree(pc.i);
        dfree(pc.x);
    }
    if (st != CSX_OK) {
        free_csc(X);
        return st;
    }
    *out = put(K_CSC, X);
    return CSX_OK;
}
