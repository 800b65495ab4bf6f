// cs_happly (csparse.py:1216-1235) for a block of right-hand sides on the device: the Q' x / Q x step of cs_qrsol
// (:1896, :1908) between the row permutation and the triangular solve.
//
// Applying the reflections is a sequence (reflection k changes the rows reflection k + 1 reads), and inside one
// reflection the reference sums v' x entry by entry.  Right-hand sides are independent, so one lane runs the
// reference's loops for one right-hand side -- the dot product accumulated in storage order, multiply and add rounded
// separately, then tau *= beta and x(i) -= v(i) tau -- over X stored row-major (the layout of every block of vectors
// here: row r of all right-hand sides is contiguous, so the 64 lanes of a wave read and write one 512-byte piece per
// entry of v, and v itself is read through the scalar path).  Bit-identical to cs_happly called reflection by
// reflection on each column.
#include "csx_internal.h"

namespace csx {

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void k_happly(int32_t n, const int32_t *__restrict__ Vp, const int32_t *__restrict__ Vi,
                                               const double *__restrict__ Vx, const double *__restrict__ beta, double *X,
                                               int32_t nrhs, int transpose) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= nrhs) return;
    for (int32_t t = 0; t < n; t++) {
        const int32_t k = transpose ? t : n - 1 - t;
        const int32_t pb = Vp[k], pe = Vp[k + 1];
        double tau = 0.0;
        for (int32_t p = pb; p < pe; p++) {            // :1229-1230
            const double prod = Vx[p] * X[(int64_t)Vi[p] * nrhs + r];
            tau = tau + prod;
        }
        tau = tau * beta[k];                           // :1231
        for (int32_t p = pb; p < pe; p++) {            // :1232-1233
            const double prod = Vx[p] * tau;
            X[(int64_t)Vi[p] * nrhs + r] = X[(int64_t)Vi[p] * nrhs + r] - prod;
        }
    }
}
#pragma clang fp contract(fast)

}  // namespace csx

using namespace csx;

extern "C" int csx_happly(csx_handle_t hV, csx_handle_t hbeta, csx_handle_t hX, int32_t nrhs, int transpose) {
    CSX_TRY(require_ready());
    Csc *V = csc(hV);
    Vec *b = vec(hbeta), *x = vec(hX);
    if (!V || !V->x || !b || !x || nrhs < 0 || b->len < V->n || x->len < (int64_t)V->m * nrhs) return CSX_EINVAL;
    if (nrhs == 0 || V->n == 0) return CSX_OK;
    hipLaunchKernelGGL(k_happly, dim3((unsigned)((nrhs + 63) / 64)), dim3(64), 0, ctx().stream, V->n, V->p, V->i, V->x,
                       (const double *)b->d, (double *)x->d, nrhs, transpose);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}
