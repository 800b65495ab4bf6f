// cs_gaxpy (csparse.py:1199-1213): y += A x, A in CSC.
//
// The reference walks the columns and scatters  y[Ai[p]] += Ax[p] * x[j]
// (:1211-1212).  A scatter needs atomics on the device; the plans below turn it
// into gathers over a cached, stably transposed copy of A ("rows": the entries
// of row r in ascending (column, position) order = the order in which the
// reference adds them into y[r]).
//
//   EXACT   one thread per row, the reference's summation order, multiply and
//           add rounded separately (no FMA): bit-identical y.
//   WAVE    one wavefront (or a 4/8/16/32-lane group) per row: coalesced loads
//           of idx[]/val[], gathered x, shuffle tree reduction.  Deterministic,
//           differs from the reference only by summation order.
//   ATOMIC  straight CSC scatter, one wavefront per column, fp64 atomics; no
//           plan needed.  Order of additions is not deterministic.
//   TILED   see csx_gaxpy_tiled.hip (matrices whose rows share no columns).
//
// Algorithmic bytes per call: 12 nnz + 4(n+1) + 8 n + 16 m  (SURVEY.md 8d).
#include <cstdlib>

#include "csx_internal.h"

namespace csx {

int gaxpy_tiled_prepare(Csc *A);                                 // csx_gaxpy_tiled.hip
int gaxpy_tiled_run(const Csc *A, const double *x, double *y);   // csx_gaxpy_tiled.hip

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_gaxpy_exact(int32_t rows, const int32_t *__restrict__ ptr,
                                                     const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double acc = y[r];
    const int32_t e = ptr[r + 1];
    for (int32_t q = ptr[r]; q < e; q++) {
        double t = val[q] * x[idx[q]];
        acc = acc + t;
    }
    y[r] = acc;
}
#pragma clang fp contract(fast)

// G lanes per row, U rows per group and step: the first G entries of the U rows are requested
// back to back (U x 3 independent loads in flight per lane) before anything is consumed.
template <int G, int U>
__global__ __launch_bounds__(256) void k_gaxpy_rows(int32_t rows, const int32_t *__restrict__ ptr,
                                                    const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                    const double *__restrict__ x, double *__restrict__ y) {
    const int sub = threadIdx.x & (G - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    // row pointers are fetched one step ahead (they head the dependent chain ptr -> idx/val -> x)
    int32_t nb[U], ne[U];
    {
        const int64_t r0 = group * U;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
            nb[u] = r0 < rows ? ptr[r] : 0;
            ne[u] = (r0 < rows && r0 + u < rows) ? ptr[r + 1] : nb[u];
        }
    }
    for (int64_t r0 = group * U; r0 < rows; r0 += ngroups * U) {
        int32_t b[U], e[U], c[U];
        double v[U], acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            b[u] = nb[u];
            e[u] = ne[u];
        }
        {
            const int64_t rn = r0 + ngroups * U;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t r = rn + u < rows ? rn + u : rows - 1;
                nb[u] = ptr[r];
                ne[u] = rn + u < rows ? ptr[r + 1] : nb[u];
            }
        }
        // the y value this lane will update (lane u of the group owns row r0 + u): requested early
        const bool owner = sub < U && r0 + sub < rows;
        const double y_old = owner ? y[r0 + sub] : 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t q = b[u] + sub;
            const bool in = q < e[u];
            c[u] = in ? idx[q] : 0;
            v[u] = in ? val[q] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double xv = x[c[u]];
            acc[u] = b[u] + sub < e[u] ? v[u] * xv : 0.0;  // lanes past the row end contribute nothing
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            for (int32_t q = b[u] + G + sub; q < e[u]; q += G) acc[u] = fma(val[q], x[idx[q]], acc[u]);
#pragma unroll
            for (int d = G >> 1; d > 0; d >>= 1) acc[u] += __shfl_xor(acc[u], d, 64);
        }
        if (owner) {
            double a = acc[0];
#pragma unroll
            for (int u = 1; u < U; u++) a = sub == u ? acc[u] : a;
            y[r0 + sub] = y_old + a;
        }
    }
}

// Wide-load variant for rows of >= ~32 entries: a lane takes FOUR consecutive entries of a row per step
// through aligned 16-byte loads (one for the indices, two for the values), so a wave instruction moves
// 1 KiB instead of 256/512 bytes -- the load pipeline tracks instructions, and wider ones keep more bytes
// in flight (the tiled kernel's stream, built the same way, runs at 6 TB/s; the 4/8-byte version at 3.9).
// Chunks are aligned to 4 entries; entries of a chunk outside [b, e) are masked.  Reads may run up to 3
// entries past the end of idx/val: every block of the device allocator carries >= 64 bytes of slack.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2v __attribute__((ext_vector_type(2)));

template <int G, int U>
__global__ __launch_bounds__(256) void k_gaxpy_rows4(int32_t rows, const int32_t *__restrict__ ptr,
                                                     const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y) {
    const int sub = threadIdx.x & (G - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    int32_t nb[U], ne[U];
    {
        const int64_t r0 = group * U;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
            nb[u] = r0 < rows ? ptr[r] : 0;
            ne[u] = (r0 < rows && r0 + u < rows) ? ptr[r + 1] : nb[u];
        }
    }
    for (int64_t r0 = group * U; r0 < rows; r0 += ngroups * U) {
        int32_t b[U], e[U];
        i32x4 c[U];
        f64x2v v0[U], v1[U];
        double acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            b[u] = nb[u];
            e[u] = ne[u];
        }
        {
            const int64_t rn = r0 + ngroups * U;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t r = rn + u < rows ? rn + u : rows - 1;
                nb[u] = ptr[r];
                ne[u] = rn + u < rows ? ptr[r + 1] : nb[u];
            }
        }
        const bool owner = sub < U && r0 + sub < rows;
        const double y_old = owner ? y[r0 + sub] : 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t q = (b[u] & ~3) + 4 * sub;
            const bool in = q < e[u];
            const int32_t qq = in ? q : 0;
            c[u] = *reinterpret_cast<const i32x4 *>(idx + qq);
            v0[u] = *reinterpret_cast<const f64x2v *>(val + qq);
            v1[u] = *reinterpret_cast<const f64x2v *>(val + qq + 2);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t q = (b[u] & ~3) + 4 * sub;
            const bool m0 = q >= b[u] && q < e[u], m1 = q + 1 >= b[u] && q + 1 < e[u];
            const bool m2 = q + 2 >= b[u] && q + 2 < e[u], m3 = q + 3 >= b[u] && q + 3 < e[u];
            const double x0 = x[m0 ? c[u].x : 0], x1 = x[m1 ? c[u].y : 0];
            const double x2 = x[m2 ? c[u].z : 0], x3 = x[m3 ? c[u].w : 0];
            double a = m0 ? v0[u].x * x0 : 0.0;
            a = m1 ? fma(v0[u].y, x1, a) : a;
            a = m2 ? fma(v1[u].x, x2, a) : a;
            a = m3 ? fma(v1[u].y, x3, a) : a;
            acc[u] = a;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            for (int32_t q = (b[u] & ~3) + 4 * (G + sub); q < e[u]; q += 4 * G) {   // rows longer than 4 G entries
                const i32x4 cc = *reinterpret_cast<const i32x4 *>(idx + q);
                const f64x2v w0 = *reinterpret_cast<const f64x2v *>(val + q);
                const f64x2v w1 = *reinterpret_cast<const f64x2v *>(val + q + 2);
                acc[u] = fma(w0.x, x[cc.x], acc[u]);            // q >= b here; only the row end needs masks
                if (q + 1 < e[u]) acc[u] = fma(w0.y, x[cc.y], acc[u]);
                if (q + 2 < e[u]) acc[u] = fma(w1.x, x[cc.z], acc[u]);
                if (q + 3 < e[u]) acc[u] = fma(w1.y, x[cc.w], acc[u]);
            }
#pragma unroll
            for (int d = G >> 1; d > 0; d >>= 1) acc[u] += __shfl_xor(acc[u], d, 64);
        }
        if (owner) {
            double a = acc[0];
#pragma unroll
            for (int u = 1; u < U; u++) a = sub == u ? acc[u] : a;
            y[r0 + sub] = y_old + a;
        }
    }
}

__global__ __launch_bounds__(256) void k_gaxpy_atomic(int32_t n, const int32_t *__restrict__ Ap,
                                                      const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                      const double *__restrict__ x, double *y) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t j = wave; j < n; j += nwaves) {
        const int32_t b = Ap[j], e = Ap[j + 1];
        const double xj = x[j];
        for (int32_t p = b + lane; p < e; p += 64) unsafeAtomicAdd(&y[Ai[p]], Ax[p] * xj);
    }
}

// Column locality probe for CSX_GAXPY_AUTO: mean over a sample of rows of (last column - first
// column) of the row in the stable transpose (columns ascending).  A row-gather SpMV streams when
// the x values a row needs sit close together; when rows span the whole of a vector that does not
// fit an XCD's L2 the LDS-resident plan wins (csx_gaxpy_tiled.hip).
__global__ __launch_bounds__(256) void k_row_span(int32_t rows, int32_t stride, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ idx, unsigned long long *sum,
                                                  unsigned int *cnt) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (r >= rows) return;
    const int32_t b = ptr[r], e = ptr[r + 1];
    if (e - b < 2) return;
    atomicAdd(sum, (unsigned long long)(idx[e - 1] - idx[b]));
    atomicAdd(cnt, 1u);
}

static int wants_tiled(const Csc *A, bool *yes) {
    *yes = false;
    const Gather *g = A->rows;
    // small problems: x fits one XCD's L2 (every XCD gathers from all of x), or too few entries to matter
    if ((int64_t)A->n * 8 <= (4ll << 20) || A->nnz < (1 << 24) || !g) return CSX_OK;
    hipStream_t s = ctx().stream;
    unsigned long long *d = nullptr;
    CSX_TRY(dalloc(&d, 2));
    CSX_HIP(hipMemsetAsync(d, 0, 16, s));
    const int32_t stride = g->rows > (1 << 16) ? g->rows >> 16 : 1;
    const int64_t samples = ((int64_t)g->rows + stride - 1) / stride;
    hipLaunchKernelGGL(k_row_span, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, s, g->rows, stride, g->ptr,
                       g->idx, d, (unsigned int *)(d + 1));
    unsigned long long h[2] = {0, 0};
    int st = CSX_OK;
    if (hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        st = CSX_ERUNTIME;
    dfree(d);
    CSX_TRY(st);
    const unsigned int cnt = (unsigned int)(h[1] & 0xffffffffu);
    if (!cnt) return CSX_OK;
    const double mean_span_bytes = (double)h[0] / cnt * 8.0;
    *yes = mean_span_bytes > (double)(4 << 20);  // a row reaches across more of x than one XCD's L2 holds
    return CSX_OK;
}

static int run_rows(const Gather *g, int64_t nnz, const double *x, double *y) {
    hipStream_t s = ctx().stream;
    if (g->rows == 0) return CSX_OK;
    const double avg = (double)nnz / (double)g->rows;
    const int64_t cap = (int64_t)ctx().cus * 32;  // workgroups of 256: 8 per CU x 4 rounds
#define CSX_ROWS(G)                                                                                       \
    {                                                                                                     \
        int64_t blocks = (((int64_t)g->rows + RU - 1) / RU * G + 255) / 256;                              \
        if (blocks > cap) blocks = cap;                                                                   \
        hipLaunchKernelGGL((k_gaxpy_rows<G, RU>), dim3((unsigned)blocks), dim3(256), 0, s, g->rows, g->ptr, \
                           g->idx, g->val, x, y);                                                         \
    }
    // 8 rows per group in flight measured 7 % faster than 4 on G-spd (0.94 vs 1.01 ms); rows with
    // fewer than ~6 entries use 4-lane groups, which need U <= 4
    static const int ru_env = ablation_env("CSX_ROWS_U") ? std::atoi(ablation_env("CSX_ROWS_U")) : 8;
    // 16-byte loads, 4 rows per group in flight: 0.69 ms on G-spd (5.7 TB/s); 8 rows: 0.74 ms; the 4/8-byte
    // kernel: 1.02 ms.  CSX_ROWS_WIDE_U = 0 (off) | 4 | 8 for ablation.
    static const int wide_env = ablation_env("CSX_ROWS_WIDE_U") ? std::atoi(ablation_env("CSX_ROWS_WIDE_U")) : 4;
#define CSX_ROWS4(G, RU)                                                                                  \
    {                                                                                                     \
        int64_t blocks = (((int64_t)g->rows + RU - 1) / RU * G + 255) / 256;                              \
        if (blocks > cap) blocks = cap;                                                                   \
        hipLaunchKernelGGL((k_gaxpy_rows4<G, RU>), dim3((unsigned)blocks), dim3(256), 0, s, g->rows, g->ptr, \
                           g->idx, g->val, x, y);                                                         \
    }
    if (wide_env && avg > 24) {
        if (wide_env == 8) {
            if (avg > 48) CSX_ROWS4(16, 8)
            else CSX_ROWS4(8, 8)
        } else {
            if (avg > 48) CSX_ROWS4(16, 4)
            else CSX_ROWS4(8, 4)
        }
    } else if (ru_env == 8 && avg > 6) {
        constexpr int RU = 8;
        if (avg > 48) CSX_ROWS(64)
        else if (avg > 24) CSX_ROWS(32)
        else if (avg > 12) CSX_ROWS(16)
        else CSX_ROWS(8)
    } else {
        constexpr int RU = 4;
        if (avg > 48) CSX_ROWS(64)
        else if (avg > 24) CSX_ROWS(32)
        else if (avg > 12) CSX_ROWS(16)
        else if (avg > 6) CSX_ROWS(8)
        else CSX_ROWS(4)
    }
#undef CSX_ROWS
#undef CSX_ROWS4
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

int csx::gaxpy_prepare_device(Csc *A, int mode) {
    if (!A->x) return CSX_EINVAL;
    switch (mode) {
        case CSX_GAXPY_ATOMIC: return CSX_OK;
        case CSX_GAXPY_TILED: return gaxpy_tiled_prepare(A);
        case CSX_GAXPY_AUTO: {
            CSX_TRY(build_row_gather(A));
            bool tiled = false;
            CSX_TRY(wants_tiled(A, &tiled));
            return tiled ? gaxpy_tiled_prepare(A) : CSX_OK;
        }
        case CSX_GAXPY_EXACT:
        case CSX_GAXPY_WAVE: return build_row_gather(A);
        default: return CSX_EINVAL;
    }
}

extern "C" int csx_gaxpy_prepare(csx_handle_t hA, int mode) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A) return CSX_EINVAL;
    return gaxpy_prepare_device(A, mode);
}

extern "C" int csx_gaxpy_plan_info(csx_handle_t hA, int *has_rows, int *has_tiled, int *key_bytes) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A) return CSX_EINVAL;
    if (has_rows) *has_rows = A->rows ? 1 : 0;
    if (has_tiled) *has_tiled = A->tiled ? 1 : 0;
    if (key_bytes) *key_bytes = A->tiled ? (A->tiled->tile_key24 ? 3 : 4) : 0;
    return CSX_OK;
}

/* Launch shape the tiled plan picked by timing its candidates when it was built (0: 4 waves x 5 groups, 1: 2 x 10,
 * 2: 8 x 4, 3: 2 x 8; -1: not tuned -- small matrix -- the default 4 x 5) and the candidates' times (ms per pass). */
extern "C" int csx_gaxpy_plan_shape(csx_handle_t hA, int *shape, double *ms4) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->tiled) return CSX_EINVAL;
    if (shape) *shape = A->tiled->shape;
    if (ms4)
        for (int k = 0; k < 4; k++) ms4[k] = A->tiled->shape_ms[k];
    return CSX_OK;
}

// y += A x on raw device pointers (x: A->n entries, y: A->m): what csx_gaxpy and the sharded SpMV (csx_comm.hip) call.
int csx::gaxpy_device(Csc *A, const double *xd, double *yd, int mode) {
    if (!A->x) return CSX_EINVAL;
    if (A->nnz == 0 || A->m == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    if (mode == CSX_GAXPY_AUTO) {
        if (!A->tiled && !A->rows) CSX_TRY(gaxpy_prepare_device(A, CSX_GAXPY_AUTO));   // first call: probe + plan
        mode = A->tiled ? CSX_GAXPY_TILED : CSX_GAXPY_WAVE;
    }
    switch (mode) {
        case CSX_GAXPY_EXACT: {
            CSX_TRY(build_row_gather(A));
            int64_t blocks = ((int64_t)A->m + 255) / 256;
            hipLaunchKernelGGL(k_gaxpy_exact, dim3((unsigned)blocks), dim3(256), 0, s, A->rows->rows, A->rows->ptr,
                               A->rows->idx, A->rows->val, xd, yd);
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        case CSX_GAXPY_WAVE:
            CSX_TRY(build_row_gather(A));
            return run_rows(A->rows, A->nnz, xd, yd);
        case CSX_GAXPY_TILED:
            CSX_TRY(gaxpy_tiled_prepare(A));
            return gaxpy_tiled_run(A, xd, yd);
        case CSX_GAXPY_ATOMIC: {
            int64_t blocks = ((int64_t)A->n + 3) / 4;
            const int64_t cap = (int64_t)ctx().cus * 32;
            if (blocks > cap) blocks = cap;
            hipLaunchKernelGGL(k_gaxpy_atomic, dim3((unsigned)blocks), dim3(256), 0, s, A->n, A->p, A->i, A->x, xd, yd);
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        default: return CSX_EINVAL;
    }
}

extern "C" int csx_gaxpy(csx_handle_t hA, csx_handle_t hx, csx_handle_t hy, int mode) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    Vec *x = vec(hx), *y = vec(hy);
    if (!A || !x || !y || !A->x || x->len < A->n || y->len < A->m) return CSX_EINVAL;
    return gaxpy_device(A, (const double *)x->d, (double *)y->d, mode);
}

// One-shot form for host arrays (the list-based reference signature, csparse.py:1199): y[0..m) += A x.
// Uploads, runs the reference-order kernel (bit-identical y), downloads; nothing stays on the device.
extern "C" int csx_gaxpy_host(int32_t m, int32_t n, const int32_t *p, const int32_t *i, const double *x, const double *xv,
                              double *yv) {
    CSX_TRY(require_ready());
    if (!xv || !yv) return CSX_EINVAL;
    csx_handle_t hA = 0, hx = 0, hy = 0;
    int st = csx_csc_upload(m, n, p, i, x, &hA);
    if (st == CSX_OK) st = csx_vec_upload(xv, n, &hx);
    if (st == CSX_OK) st = csx_vec_upload(yv, m, &hy);
    if (st == CSX_OK) st = csx_gaxpy(hA, hx, hy, CSX_GAXPY_EXACT);
    if (st == CSX_OK) st = csx_vec_download(hy, yv, m);
    if (hA) (void)csx_free(hA);
    if (hx) (void)csx_free(hx);
    if (hy) (void)csx_free(hy);
    return st;
}
