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
#include <algorithm>
#include <vector>

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
// One level of the schedule: the reflections cols[c0 .. c1) touch disjoint rows, so they are applied side by side --
// wave = (reflection, chunk of 64 right-hand sides), lane = right-hand side, the same loops as above.  Which level a
// reflection belongs to does not change a bit of the result: a reflection only reads and writes its own rows.
__global__ __launch_bounds__(64) void k_happly_level(const int32_t *__restrict__ cols, int32_t c0, int32_t c1,
                                                     const int32_t *__restrict__ Vp, const int32_t *__restrict__ Vi,
                                                     const double *__restrict__ Vx, const double *__restrict__ beta,
                                                     double *X, int32_t nrhs, int32_t chunks) {
    const int32_t w = (int32_t)blockIdx.x;
    const int32_t ci = c0 + w / chunks, r = (w % chunks) * 64 + (int32_t)threadIdx.x;
    if (ci >= c1 || r >= nrhs) return;
    const int32_t k = cols[ci];
    const int32_t pb = Vp[k], pe = Vp[k + 1];
    double tau = 0.0;
    for (int32_t p = pb; p < pe; p++) {
        const double prod = Vx[p] * X[(int64_t)Vi[p] * nrhs + r];
        tau = tau + prod;
    }
    tau = tau * beta[k];
    for (int32_t p = pb; p < pe; p++) {
        const double prod = Vx[p] * tau;
        X[(int64_t)Vi[p] * nrhs + r] = X[(int64_t)Vi[p] * nrhs + r] - prod;
    }
}
#pragma clang fp contract(fast)

constexpr int32_t HAPPLY_MAX_LEVELS = 4096;   // beyond that (a chain of dependent reflections) one launch walks them all

static int download_i32(std::vector<int32_t> &h, const int32_t *d, size_t count) {
    h.resize(count);
    if (count) CSX_HIP(hipMemcpyAsync(h.data(), d, count * sizeof(int32_t), hipMemcpyDeviceToHost, ctx().stream));
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return CSX_OK;
}

// level[k] = 1 + the highest level among the earlier reflections that share a row with k (host, O(nnz V))
static int house_levels(Csc *V) {
    if (V->house) return CSX_OK;
    std::vector<int32_t> hp, hi;
    CSX_TRY(download_i32(hp, V->p, (size_t)V->n + 1));
    CSX_TRY(download_i32(hi, V->i, (size_t)V->nnz));
    std::vector<int32_t> rowlev((size_t)V->m, 0), level((size_t)V->n, 0);
    int32_t nlev = 0;
    for (int32_t k = 0; k < V->n; k++) {
        int32_t lv = 0;
        for (int32_t p = hp[(size_t)k]; p < hp[(size_t)k + 1]; p++) lv = std::max(lv, rowlev[(size_t)hi[(size_t)p]]);
        level[(size_t)k] = lv;                                  // 0-based
        for (int32_t p = hp[(size_t)k]; p < hp[(size_t)k + 1]; p++) rowlev[(size_t)hi[(size_t)p]] = lv + 1;
        nlev = std::max(nlev, lv + 1);
    }
    HouseLevels *H = new HouseLevels();
    H->nlevels = nlev;
    H->ptr.assign((size_t)nlev + 1, 0);
    for (int32_t k = 0; k < V->n; k++) H->ptr[(size_t)level[(size_t)k] + 1]++;
    for (int32_t l = 0; l < nlev; l++) H->ptr[(size_t)l + 1] += H->ptr[(size_t)l];
    std::vector<int32_t> fill(H->ptr.begin(), H->ptr.end() - 1), cols((size_t)V->n);
    for (int32_t k = 0; k < V->n; k++) cols[(size_t)fill[(size_t)level[(size_t)k]]++] = k;
    int st = dalloc(&H->cols, (size_t)std::max<int32_t>(V->n, 1));
    if (st == CSX_OK && V->n > 0 &&
        (hipMemcpyAsync(H->cols, cols.data(), (size_t)V->n * sizeof(int32_t), hipMemcpyHostToDevice, ctx().stream) != hipSuccess ||
         hipStreamSynchronize(ctx().stream) != hipSuccess))
        st = CSX_ERUNTIME;
    if (st != CSX_OK) {
        dfree(H->cols);
        delete H;
        return st;
    }
    V->house = H;
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_happly(csx_handle_t hV, csx_handle_t hbeta, csx_handle_t hX, int32_t nrhs, int transpose) {
    CSX_TRY(require_ready());
    Csc *V = csc(hV);
    Vec *b = vec(hbeta), *x = vec(hX);
    if (!V || !V->x || !b || !x || nrhs < 0 || b->len < V->n || x->len < (int64_t)V->m * nrhs) return CSX_EINVAL;
    if (nrhs == 0 || V->n == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    CSX_TRY(house_levels(V));
    const HouseLevels *H = V->house;
    if (H->nlevels > HAPPLY_MAX_LEVELS || H->nlevels == V->n) {   // nothing to run side by side: one launch
        hipLaunchKernelGGL(k_happly, dim3((unsigned)((nrhs + 63) / 64)), dim3(64), 0, s, V->n, V->p, V->i, V->x,
                           (const double *)b->d, (double *)x->d, nrhs, transpose);
        CSX_LAUNCH_CHECK();
        return CSX_OK;
    }
    const int32_t chunks = (nrhs + 63) / 64;
    for (int32_t t = 0; t < H->nlevels; t++) {
        const int32_t l = transpose ? t : H->nlevels - 1 - t;   // Q' x: reflections ascending; Q x: descending
        const int32_t c0 = H->ptr[(size_t)l], c1 = H->ptr[(size_t)l + 1];
        hipLaunchKernelGGL(k_happly_level, dim3((unsigned)((int64_t)(c1 - c0) * chunks)), dim3(64), 0, s, H->cols, c0, c1, V->p,
                           V->i, V->x, (const double *)b->d, (double *)x->d, nrhs, chunks);
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}
