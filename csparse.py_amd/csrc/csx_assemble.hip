// Assembly and reshaping around the hot path: the callers either side of
// cs_multiply / cs_transpose that are the same device primitives in other clothes.
//
//   cs_compress (csparse.py:647-673)  triplets -> CSC: the reference's counting sort by column, stable in
//                                     triplet order = one stable radix sort by column key (csx_sort.hip)
//   cs_add      (csparse.py:163-192)  alpha A + beta B, column pattern in first-touch order over
//                                     [A(:,j) then B(:,j)]:  C = [A B] * [alpha I; beta I]  through the
//                                     SpGEMM kernels, whose column order IS the reference's cs_scatter order
//   cs_dupl     (csparse.py:1035-1065) sum duplicates, first occurrence keeps its place:  C = A * I
//   cs_dropzeros / cs_droptol (csparse.py:1019-1031, 1002-1014; cs_fkeep :1172-1196)  order-preserving
//                                     compaction: per-column counts (ballots), scan, fill
//   cs_permute  (csparse.py:1666-1693) C = P A Q: columns gathered by q, rows renamed by pinv, order kept
//   cs_symperm  (csparse.py:2220-2255) upper triangle of P A P': entries i <= j renamed, bucketed by their
//                                     new column max(i2, j2) in (j, p) order = filter + stable radix sort
//
// p[] and i[] of every result are bit-identical to the reference's; x[] too wherever a value is copied or
// is the sum of at most two terms (cs_add of matrices without duplicates), otherwise sums are in arrival
// order and agree to rounding.
#include <cstring>
#include <vector>

#include "csx_internal.h"

namespace csx {

int multiply_device(const Csc *A, const Csc *B, Csc *C);   // csx_spgemm.hip

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

static int upload_i32(int32_t **d, const int32_t *h, size_t count) {
    CSX_TRY(dalloc(d, count));
    if (count) CSX_HIP(hipMemcpyAsync(*d, h, count * sizeof(int32_t), hipMemcpyHostToDevice, ctx().stream));
    return CSX_OK;
}

static bool is_permutation(const int32_t *p, int32_t n) {
    std::vector<char> seen((size_t)n, 0);
    for (int32_t k = 0; k < n; k++) {
        if (p[k] < 0 || p[k] >= n || seen[(size_t)p[k]]) return false;
        seen[(size_t)p[k]] = 1;
    }
    return true;
}

// ---- cs_add / cs_dupl helpers --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cat_ptr(int32_t n, const int32_t *__restrict__ Ap, const int32_t *__restrict__ Bp,
                                                 int32_t *__restrict__ p) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= n) p[j] = Ap[j];
    if (j >= 1 && j <= n) p[n + j] = Ap[n] + Bp[j];
}

// S = [alpha I; beta I]  (2n x n), or the n x n identity when stacked == 0
__global__ __launch_bounds__(256) void k_selector(int32_t n, int stacked, double alpha, double beta, int32_t *__restrict__ p,
                                                  int32_t *__restrict__ i, double *__restrict__ x) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    const int per = stacked ? 2 : 1;
    p[j] = (int32_t)(per * j);
    if (j == n) return;
    i[per * j] = (int32_t)j;
    if (x) x[per * j] = alpha;
    if (stacked) {
        i[2 * j + 1] = (int32_t)(n + j);
        if (x) x[2 * j + 1] = beta;
    }
}

static void drop_fields(Csc *M) {
    dfree(M->p);
    dfree(M->i);
    dfree(M->x);
    M->p = M->i = nullptr;
    M->x = nullptr;
}

static int add_device(const Csc *A, const Csc *B, double alpha, double beta, Csc *C) {
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    const bool values = A->x && B->x;
    const int64_t tot = (int64_t)A->nnz + B->nnz;
    if (tot > 0x7FFFFFFFll || 2 * (int64_t)n > 0x7FFFFFFFll) return CSX_EINVAL;
    Csc cat, S;
    cat.m = A->m;
    cat.n = 2 * n;
    cat.nnz = (int32_t)tot;
    S.m = 2 * n;
    S.n = n;
    S.nnz = 2 * n;
    int st = dalloc(&cat.p, (size_t)2 * n + 1);
    if (st == CSX_OK) st = dalloc(&cat.i, (size_t)tot);
    if (st == CSX_OK && values) st = dalloc(&cat.x, (size_t)tot);
    if (st == CSX_OK) st = dalloc(&S.p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&S.i, (size_t)2 * n);
    if (st == CSX_OK && values) st = dalloc(&S.x, (size_t)2 * n);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_cat_ptr, dim3(blocks_for((int64_t)n + 1)), dim3(256), 0, s, n, A->p, B->p, cat.p);
        hipLaunchKernelGGL(k_selector, dim3(blocks_for((int64_t)n + 1)), dim3(256), 0, s, n, 1, alpha, beta, S.p, S.i, S.x);
        hipError_t e = hipSuccess;
        if (A->nnz) e = hipMemcpyAsync(cat.i, A->i, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && B->nnz) e = hipMemcpyAsync(cat.i + A->nnz, B->i, (size_t)B->nnz * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && values && A->nnz) e = hipMemcpyAsync(cat.x, A->x, (size_t)A->nnz * 8, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && values && B->nnz)
            e = hipMemcpyAsync(cat.x + A->nnz, B->x, (size_t)B->nnz * 8, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) st = CSX_ERUNTIME;
    }
    if (st == CSX_OK) st = multiply_device(&cat, &S, C);
    (void)hipStreamSynchronize(s);
    drop_fields(&cat);
    drop_fields(&S);
    return st;
}

static int dupl_device(const Csc *A, Csc *C) {
    hipStream_t s = ctx().stream;
    Csc I;
    I.m = I.n = A->n;
    I.nnz = A->n;
    int st = dalloc(&I.p, (size_t)A->n + 1);
    if (st == CSX_OK) st = dalloc(&I.i, (size_t)A->n);
    if (st == CSX_OK && A->x) st = dalloc(&I.x, (size_t)A->n);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_selector, dim3(blocks_for((int64_t)A->n + 1)), dim3(256), 0, s, A->n, 0, 1.0, 0.0, I.p, I.i, I.x);
        st = multiply_device(A, &I, C);
    }
    (void)hipStreamSynchronize(s);
    drop_fields(&I);
    return st;
}

// ---- order-preserving filters --------------------------------------------------------------------------
enum { KEEP_NONZERO = 0, KEEP_ABOVE_TOL = 1, KEEP_UPPER = 2 };

__device__ __forceinline__ bool keep_entry(int mode, double tol, int32_t i, int32_t j, double a) {
    if (mode == KEEP_NONZERO) return a != 0;
    if (mode == KEEP_ABOVE_TOL) return fabs(a) > tol;
    return i <= j;
}

// one wave per column; FILL = false: cnt[j] = kept entries; FILL = true: write them at optr[j] in order.
// For KEEP_UPPER the outputs are the renamed (row, column-key) pairs of cs_symperm instead of (row, value).
template <bool FILL>
__global__ __launch_bounds__(256) void k_filter(int32_t n, int mode, double tol, const int32_t *__restrict__ Ap,
                                                const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                const int32_t *__restrict__ pinv, int32_t *__restrict__ cnt,
                                                const int32_t *__restrict__ optr, int32_t *__restrict__ oi,
                                                double *__restrict__ ox, uint32_t *__restrict__ okey) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t b = Ap[j], e = Ap[j + 1];
    int32_t run = FILL ? optr[j] : 0;
    const int32_t j2 = (mode == KEEP_UPPER && pinv) ? pinv[j] : (int32_t)j;
    for (int32_t p0 = b; p0 < e; p0 += 64) {
        const int32_t p = p0 + lane;
        bool keep = false;
        int32_t i = 0;
        double a = 1.0;
        if (p < e) {
            i = Ai[p];
            if (Ax) a = Ax[p];
            keep = keep_entry(mode, tol, i, (int32_t)j, a);
        }
        const unsigned long long bal = __ballot(keep);
        if (FILL && keep) {
            const int32_t q = run + __popcll(bal & ((1ull << lane) - 1ull));
            if (mode == KEEP_UPPER) {
                const int32_t i2 = pinv ? pinv[i] : i;
                oi[q] = i2 < j2 ? i2 : j2;
                okey[q] = (uint32_t)(i2 > j2 ? i2 : j2);
            } else {
                oi[q] = i;
            }
            if (ox) ox[q] = a;
        }
        run += __popcll(bal);
    }
    if (!FILL && lane == 0) cnt[j] = run;
}

static int drop_device(const Csc *A, int mode, double tol, Csc *C) {
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    C->m = A->m;
    C->n = n;
    C->owns = true;
    int32_t *cnt = nullptr;
    int64_t total = 0;
    int st = dalloc(&cnt, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&C->p, (size_t)n + 1);
    if (st == CSX_OK) {
        if (n) hipLaunchKernelGGL(k_filter<false>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, mode, tol, A->p, A->i, A->x,
                                  nullptr, cnt, nullptr, nullptr, nullptr, nullptr);
        st = scan_exclusive_i32(cnt, C->p, n, &total);
    }
    if (st == CSX_OK) {
        C->nnz = (int32_t)total;
        st = dalloc(&C->i, (size_t)total);
        if (st == CSX_OK && A->x) st = dalloc(&C->x, (size_t)total);
    }
    if (st == CSX_OK && n)
        hipLaunchKernelGGL(k_filter<true>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, mode, tol, A->p, A->i, A->x, nullptr,
                           nullptr, C->p, C->i, C->x, nullptr);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    dfree(cnt);
    return st;
}

static int symperm_device(const Csc *A, const int32_t *pinv_h, bool values, Csc *C) {
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    const bool with_values = values && A->x;
    C->m = C->n = n;
    C->owns = true;
    int32_t *pinv = nullptr, *cnt = nullptr, *optr = nullptr, *ri = nullptr;
    uint32_t *key = nullptr, *skey = nullptr;
    double *rx = nullptr;
    int64_t total = 0;
    int st = CSX_OK;
    if (pinv_h) st = upload_i32(&pinv, pinv_h, (size_t)n);
    if (st == CSX_OK) st = dalloc(&cnt, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&optr, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&C->p, (size_t)n + 1);
    if (st == CSX_OK) {
        if (n) hipLaunchKernelGGL(k_filter<false>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, (int)KEEP_UPPER, 0.0, A->p, A->i,
                                  nullptr, pinv, cnt, nullptr, nullptr, nullptr, nullptr);
        st = scan_exclusive_i32(cnt, optr, n, &total);
    }
    if (st == CSX_OK) {
        C->nnz = (int32_t)total;
        st = dalloc(&C->i, (size_t)total);
        if (st == CSX_OK && with_values) st = dalloc(&C->x, (size_t)total);
        if (st == CSX_OK) st = dalloc(&ri, (size_t)total);
        if (st == CSX_OK) st = dalloc(&key, (size_t)total);
        if (st == CSX_OK) st = dalloc(&skey, (size_t)total);
        if (st == CSX_OK && with_values) st = dalloc(&rx, (size_t)total);
    }
    if (st == CSX_OK && total > 0) {
        hipLaunchKernelGGL(k_filter<true>, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, (int)KEEP_UPPER, 0.0, A->p, A->i,
                           with_values ? A->x : nullptr, pinv, nullptr, optr, ri, rx, key);
        st = stable_sort_by_key(key, (const uint32_t *)ri, rx, total, (uint32_t)n, skey, (uint32_t *)C->i, C->x);
        if (st == CSX_OK) st = boundaries_from_sorted(skey, total, n, C->p);
    } else if (st == CSX_OK) {
        if (hipMemsetAsync(C->p, 0, ((size_t)n + 1) * sizeof(int32_t), s) != hipSuccess) st = CSX_ERUNTIME;
    }
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    for (void *q : {(void *)pinv, (void *)cnt, (void *)optr, (void *)ri, (void *)key, (void *)skey, (void *)rx}) dfree(q);
    return st;
}

// ---- cs_permute ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_perm_len(int32_t n, const int32_t *__restrict__ Ap, const int32_t *__restrict__ q,
                                                  int32_t *__restrict__ len) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t j = q ? q[k] : (int32_t)k;
    len[k] = Ap[j + 1] - Ap[j];
}

__global__ __launch_bounds__(256) void k_perm_fill(int32_t n, const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                   const double *__restrict__ Ax, const int32_t *__restrict__ pinv,
                                                   const int32_t *__restrict__ q, const int32_t *__restrict__ Cp,
                                                   int32_t *__restrict__ Ci, double *__restrict__ Cx) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= n) return;
    const int32_t j = q ? q[k] : (int32_t)k;
    const int32_t b = Ap[j], len = Ap[j + 1] - b, o = Cp[k];
    for (int32_t t = lane; t < len; t += 64) {
        const int32_t i = Ai[b + t];
        Ci[o + t] = pinv ? pinv[i] : i;
        if (Cx) Cx[o + t] = Ax[b + t];
    }
}

static int permute_device(const Csc *A, const int32_t *pinv_h, const int32_t *q_h, bool values, Csc *C) {
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    const bool with_values = values && A->x;
    C->m = A->m;
    C->n = n;
    C->nnz = A->nnz;
    C->owns = true;
    int32_t *pinv = nullptr, *q = nullptr, *len = nullptr;
    int st = CSX_OK;
    if (pinv_h) st = upload_i32(&pinv, pinv_h, (size_t)A->m);
    if (st == CSX_OK && q_h) st = upload_i32(&q, q_h, (size_t)n);
    if (st == CSX_OK) st = dalloc(&len, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&C->p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&C->i, (size_t)A->nnz);
    if (st == CSX_OK && with_values) st = dalloc(&C->x, (size_t)A->nnz);
    if (st == CSX_OK) {
        if (n) hipLaunchKernelGGL(k_perm_len, dim3(blocks_for(n)), dim3(256), 0, s, n, A->p, q, len);
        st = scan_exclusive_i32(len, C->p, n, nullptr);
    }
    if (st == CSX_OK && n)
        hipLaunchKernelGGL(k_perm_fill, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, A->p, A->i, with_values ? A->x : nullptr,
                           pinv, q, C->p, C->i, C->x);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    dfree(pinv);
    dfree(q);
    dfree(len);
    return st;
}

// ---- cs_compress ---------------------------------------------------------------------------------------
static int compress_device(int32_t m, int32_t n, int64_t nz, const int32_t *Ti, const int32_t *Tj, const double *Tx, Csc *C) {
    hipStream_t s = ctx().stream;
    C->m = m;
    C->n = n;
    C->nnz = (int32_t)nz;
    C->owns = true;
    int32_t *di = nullptr, *dj = nullptr;
    uint32_t *skey = nullptr;
    double *dx = nullptr;
    int st = dalloc(&C->p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&C->i, (size_t)nz);
    if (st == CSX_OK && Tx) st = dalloc(&C->x, (size_t)nz);
    if (st == CSX_OK && nz == 0) {
        if (hipMemsetAsync(C->p, 0, ((size_t)n + 1) * sizeof(int32_t), s) != hipSuccess) st = CSX_ERUNTIME;
        if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
        return st;
    }
    if (st == CSX_OK) st = upload_i32(&di, Ti, (size_t)nz);
    if (st == CSX_OK) st = upload_i32(&dj, Tj, (size_t)nz);
    if (st == CSX_OK && Tx) {
        st = dalloc(&dx, (size_t)nz);
        if (st == CSX_OK && hipMemcpyAsync(dx, Tx, (size_t)nz * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    if (st == CSX_OK) st = dalloc(&skey, (size_t)nz);
    if (st == CSX_OK)
        st = stable_sort_by_key((const uint32_t *)dj, (const uint32_t *)di, dx, nz, (uint32_t)n, skey, (uint32_t *)C->i, C->x);
    if (st == CSX_OK) st = boundaries_from_sorted(skey, nz, n, C->p);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    dfree(di);
    dfree(dj);
    dfree(dx);
    dfree(skey);
    return st;
}

// ---- column block (sharding a matrix by columns) ---------------------------------------------
__global__ __launch_bounds__(256) void k_shift_ptr(int32_t count, const int32_t *__restrict__ Ap, int32_t first,
                                                   int32_t *__restrict__ p) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= count) p[j] = Ap[first + j] - Ap[first];
}

int col_block_device(const Csc *A, int32_t first, int32_t count, Csc *C) {
    hipStream_t s = ctx().stream;
    int32_t ends[2] = {0, 0};
    CSX_HIP(hipMemcpyAsync(&ends[0], A->p + first, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipMemcpyAsync(&ends[1], A->p + first + count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    const int32_t nnz = ends[1] - ends[0];
    C->m = A->m;
    C->n = count;
    C->nnz = nnz;
    C->owns = true;
    CSX_TRY(dalloc(&C->p, (size_t)count + 1));
    CSX_TRY(dalloc(&C->i, (size_t)nnz));
    if (A->x) CSX_TRY(dalloc(&C->x, (size_t)nnz));
    hipLaunchKernelGGL(k_shift_ptr, dim3(blocks_for((int64_t)count + 1)), dim3(256), 0, s, count, A->p, first, C->p);
    if (nnz) CSX_HIP(hipMemcpyAsync(C->i, A->i + ends[0], (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (nnz && A->x) CSX_HIP(hipMemcpyAsync(C->x, A->x + ends[0], (size_t)nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    CSX_HIP(hipStreamSynchronize(s));
    return CSX_OK;
}

// ---- cs_norm (csparse.py:1647-1663): largest column sum of |a| -------------------------------------------
// One thread per column, entries added in storage order: every column sum, hence the maximum, has the
// reference's bits.  Non-negative doubles order like their bit patterns, so the maximum is an integer atomic.
__global__ __launch_bounds__(256) void k_norm1(int32_t n, const int32_t *__restrict__ Ap, const double *__restrict__ Ax,
                                               unsigned long long *best) {
#pragma clang fp contract(off)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double sum = 0.0;
    for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) sum += fabs(Ax[p]);
    if (sum == sum) atomicMax(best, (unsigned long long)__double_as_longlong(sum));
    else atomicMax(best + 1, 1ull);   // a NaN column: max(best, NaN) in the reference keeps `best`; remembered only for the caller
}

template <class F>
static int make_csc(csx_handle_t *out, F &&build) {
    Csc *C = new Csc();
    const int st = build(C);
    if (st != CSX_OK) {
        free_csc(C);
        return st;
    }
    *out = put(K_CSC, C);
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_compress(int32_t m, int32_t n, int64_t nz, const int32_t *Ti, const int32_t *Tj, const double *Tx,
                            csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (m < 0 || n < 0 || nz < 0 || nz > 0x7FFFFFFFll || !out || (nz > 0 && (!Ti || !Tj))) return CSX_EINVAL;
    for (int64_t k = 0; k < nz; k++)
        if (Tj[k] < 0 || Tj[k] >= n || Ti[k] < 0 || Ti[k] >= m) return CSX_EINVAL;   // the reference would raise IndexError
    return make_csc(out, [&](Csc *C) { return compress_device(m, n, nz, Ti, Tj, Tx, C); });
}

extern "C" int csx_add(csx_handle_t hA, csx_handle_t hB, double alpha, double beta, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA), *B = csc(hB);
    if (!A || !B || !out || A->m != B->m || A->n != B->n) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return add_device(A, B, alpha, beta, C); });
}

extern "C" int csx_dupl(csx_handle_t hA, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return dupl_device(A, C); });
}

extern "C" int csx_drop(csx_handle_t hA, int mode, double tol, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out || !A->x || (mode != KEEP_NONZERO && mode != KEEP_ABOVE_TOL)) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return drop_device(A, mode, tol, C); });
}

extern "C" int csx_permute(csx_handle_t hA, const int32_t *pinv, const int32_t *q, int values, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out) return CSX_EINVAL;
    if (pinv)
        for (int32_t k = 0; k < A->m; k++)
            if (pinv[k] < 0) return CSX_EINVAL;
    if (q)
        for (int32_t k = 0; k < A->n; k++)
            if (q[k] < 0 || q[k] >= A->n) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return permute_device(A, pinv, q, values != 0, C); });
}

extern "C" int csx_symperm(csx_handle_t hA, const int32_t *pinv, int values, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out || A->m != A->n) return CSX_EINVAL;
    if (pinv && !is_permutation(pinv, A->n)) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return symperm_device(A, pinv, values != 0, C); });
}

extern "C" int csx_csc_col_block(csx_handle_t hA, int32_t first, int32_t count, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out || first < 0 || count < 0 || (int64_t)first + count > A->n) return CSX_EINVAL;
    return make_csc(out, [&](Csc *C) { return col_block_device(A, first, count, C); });
}

extern "C" int csx_norm1(csx_handle_t hA, double *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out || !A->x) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    unsigned long long *d = nullptr, h[2] = {0, 0};
    CSX_TRY(dalloc(&d, 2));
    int st = CSX_OK;
    if (hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) st = CSX_ERUNTIME;
    if (st == CSX_OK && A->n > 0) hipLaunchKernelGGL(k_norm1, dim3(blocks_for(A->n)), dim3(256), 0, s, A->n, A->p, A->x, d);
    if (st == CSX_OK && (hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
        st = CSX_ERUNTIME;
    dfree(d);
    CSX_TRY(st);
    double v;
    std::memcpy(&v, &h[0], sizeof v);
    *out = v;
    return CSX_OK;
}
