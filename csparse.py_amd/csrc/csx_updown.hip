// cs_updown (csparse.py:2318-2365): sparse Cholesky rank-1 update / downdate, L L' + sigma w w', in place.
//
// The reference walks the elimination-tree path from f = min(find(w)) to the root; for every column j on the
// path a handful of scalars (alpha, beta2, delta, gamma) and then an update of w and of column j of L, entry
// by entry.  The scalars form a recurrence along the path (beta), so the path is sequential; a column's
// entries are independent of each other (their rows are distinct).  One workgroup walks the path, every thread
// computes the column's scalars for itself (same inputs, same operations: no broadcast needed), the threads
// share the column's entries, and a workgroup barrier separates columns (w(parent j) is written by this column).
// Every operation is the reference's, multiply and add rounded separately: L.x comes out bit-identical, also
// when a downdate stops part way because L L' - w w' is not positive definite.
#include "csx_internal.h"

namespace csx {

#pragma clang fp contract(off)
__global__ __launch_bounds__(1024) void k_updown(int32_t n, const int32_t *__restrict__ Lp,
                                                 const int32_t *__restrict__ Li, double *Lx,
                                                 const int32_t *__restrict__ parent, int32_t f, double sigma, double *w,
                                                 int *result) {
    __shared__ double s_wj, s_ljj;
    double beta = 1.0, beta2 = 1.0;
    int32_t j = f;
    for (int32_t steps = 0; j != -1 && steps <= n; steps++) {
        const int32_t p = Lp[j], e = Lp[j + 1];
        if (threadIdx.x == 0) {
            s_wj = w[j];
            s_ljj = Lx[p];
        }
        __syncthreads();
        const double wj = s_wj, ljj = s_ljj;
        const double alpha = wj / ljj;
        const double sa = sigma * alpha;
        beta2 = beta * beta + sa * alpha;
        if (beta2 <= 0.0) break;                          // not positive definite (uniform: every thread sees it)
        beta2 = sqrt(beta2);
        const double delta = sigma > 0.0 ? beta / beta2 : beta2 / beta;
        const double gamma = sa / (beta2 * beta);
        if (threadIdx.x == 0) {
            const double dl = delta * ljj;
            Lx[p] = sigma > 0.0 ? dl + gamma * wj : dl + 0.0;
        }
        beta = beta2;
        for (int32_t q = p + 1 + threadIdx.x; q < e; q += blockDim.x) {
            const int32_t r = Li[q];
            const double lx = Lx[q];
            const double w1 = w[r];
            const double w2 = w1 - alpha * lx;
            w[r] = w2;
            Lx[q] = delta * lx + gamma * (sigma > 0.0 ? w1 : w2);
        }
        j = parent[j];
        __syncthreads();                                  // w(parent) and the LDS scalars are reused next round
    }
    if (threadIdx.x == 0) result[0] = beta2 > 0.0 ? 1 : 0;
}
#pragma clang fp contract(fast)

}  // namespace csx

using namespace csx;

extern "C" int csx_updown(csx_handle_t hL, int sigma, int32_t cnz, const int32_t *Ci, const double *Cx,
                          const int32_t *parent, int *ok) {
    CSX_TRY(require_ready());
    Csc *L = csc(hL);
    if (!L || !L->x || L->m != L->n || !parent || !ok || cnz < 0 || (cnz > 0 && (!Ci || !Cx)) || (sigma != 1 && sigma != -1))
        return CSX_EINVAL;
    *ok = 1;
    const int32_t n = L->n;
    if (cnz == 0 || n == 0) return CSX_OK;               // "return if C empty"
    int32_t f = Ci[0];
    for (int32_t q = 0; q < cnz; q++) {
        if (Ci[q] < 0 || Ci[q] >= n) return CSX_EINVAL;
        f = Ci[q] < f ? Ci[q] : f;
    }
    for (int32_t j = 0; j < n; j++)
        if (parent[j] < -1 || parent[j] >= n) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    double *w = nullptr;
    int32_t *d_parent = nullptr;
    int *d_res = nullptr;
    CSX_TRY(tmp.alloc(&w, (size_t)n));
    CSX_TRY(tmp.alloc(&d_parent, (size_t)n));
    CSX_TRY(tmp.alloc(&d_res, 1));
    // w = 0 everywhere (the reference's xalloc), then w = C; later duplicates of a row win, as in the reference
    std::vector<double> hw((size_t)n, 0.0);
    for (int32_t q = 0; q < cnz; q++) hw[(size_t)Ci[q]] = Cx[q];
    CSX_HIP(hipMemcpyAsync(w, hw.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(d_parent, parent, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_updown, dim3(1), dim3(1024), 0, s, n, L->p, L->i, L->x, d_parent, f, (double)sigma, w, d_res);
    CSX_LAUNCH_CHECK();
    int res = 0;
    CSX_HIP(hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    *ok = res;
    // the factor's values changed: plans cached on it are stale
    free_gather(L->rows);
    L->rows = nullptr;
    free_tiled(L->tiled);
    L->tiled = nullptr;
    return CSX_OK;
}
