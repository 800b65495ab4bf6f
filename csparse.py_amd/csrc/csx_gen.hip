// Synthetic inputs of the benchmark configs, generated on
// the device from a counter-based hash so that the host (tests/synth.py, numpy)
// reproduces them bit for bit without shipping arrays.
//
//   u64  mix(z)      : splitmix64 finaliser
//   u64  H(seed, c)  : mix(seed ^ mix(c))
//   f64  U(seed, c)  : (H >> 11) * 2^-53            in [0, 1)
//
// G-rand (headline SpMV): n-by-n, `per_col` entries in every column; entry k of
// column j sits in row  k*W + H(seed, j*per_col+k) % W_k  where W = n / per_col
// is the stratum height (the last stratum takes the remainder), so the rows of a
// column are distinct and ascending and uniformly spread over [0, n).
// Values 0.5 + U(seed+1, j*per_col+k).
//
// G-rand, uniform draw (SURVEY 8d to the letter): column j holds `per_col` (<= 64) DISTINCT rows drawn
// uniformly from [0, n), stored ascending.  One wavefront per column, lane t draws
// r = H(seed, (j*per_col + t)*64 + a) % n with attempt a = 0; the lanes are sorted by (r, t) (bitonic, wave
// shuffles); a lane whose r equals its lower neighbour's redraws with a + 1; repeat until all differ.
// Values by position as above.  tests/synth.py:grand_uniform is the numpy twin.
//
// G-spd (Cholesky / batched solves): block diagonal, `nblocks` dense bs-by-bs
// blocks  B = R R' / bs + bs I,  R[r][k] = -1 + 2 U(seed, (b*bs + r)*bs + k),
// the sum over k taken in ascending k with separately rounded multiply and add,
// so B is bitwise symmetric.  Rows ascending inside every column.
//
// Right-hand sides: B[i][r] = 1 + (i + col0 + r) / n  (csparse_test.py:123-127,
// shifted per column).
#include "csx_internal.h"

namespace csx {

#pragma clang fp contract(off)

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash2(uint64_t seed, uint64_t c) { return mix64(seed ^ mix64(c)); }
__host__ __device__ __forceinline__ double unit(uint64_t h) { return (double)(h >> 11) * 0x1.0p-53; }

__global__ __launch_bounds__(256) void k_gen_grand(int32_t n, int32_t per_col, uint64_t seed, int32_t *Ap, int32_t *Ai,
                                                   double *Ax) {
    const int64_t nnz = (int64_t)n * per_col;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n) Ap[e] = (int32_t)(e * per_col);
    if (e >= nnz) return;
    const int32_t k = (int32_t)(e % per_col);
    const int32_t W = n / per_col;
    const int32_t Wk = (k == per_col - 1) ? n - (per_col - 1) * W : W;
    Ai[e] = k * W + (int32_t)(hash2(seed, (uint64_t)e) % (uint64_t)Wk);
    Ax[e] = 0.5 + unit(hash2(seed + 1, (uint64_t)e));
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void k_gen_grand_uniform(int32_t n, int32_t per_col, uint64_t seed, int32_t *Ap,
                                                           int32_t *Ai, double *Ax) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per column
    if (j > n) return;
    if (lane == 0) Ap[j] = (int32_t)(j * per_col);
    if (j == n) return;
    const uint64_t e = (uint64_t)j * per_col + lane;
    // key = row << 12 | drawing lane << 6 | attempt; lanes beyond per_col hold the maximum and sort last
    uint64_t key = ~0ull;
    if (lane < per_col) key = ((hash2(seed, e * 64) % (uint64_t)n) << 12) | ((uint64_t)lane << 6);
    for (int round = 0; round < 64; round++) {
        for (int k = 2; k <= 64; k <<= 1)
            for (int d = k >> 1; d > 0; d >>= 1) {
                const uint64_t other = shfl_xor_u64(key, d);
                const bool up = (lane & k) == 0, low = (lane & d) == 0;
                const uint64_t mn = key < other ? key : other, mx = key < other ? other : key;
                key = (up == low) ? mn : mx;
            }
        const uint32_t blo = (uint32_t)__shfl_up((int)(uint32_t)key, 1, 64);
        const uint32_t bhi = (uint32_t)__shfl_up((int)(uint32_t)(key >> 32), 1, 64);
        const uint64_t below = ((uint64_t)bhi << 32) | blo;  // the sorted neighbour one lane down
        const bool dup = lane > 0 && lane < per_col && (key >> 12) == (below >> 12);
        if (__ballot(dup) == 0ull) break;
        if (dup) {
            const uint64_t t = (key >> 6) & 63, a = (key & 63) + 1;
            const uint64_t et = (uint64_t)j * per_col + t;
            key = ((hash2(seed, et * 64 + a) % (uint64_t)n) << 12) | (t << 6) | a;
        }
    }
    if (lane < per_col) {
        Ai[e] = (int32_t)(key >> 12);
        Ax[e] = 0.5 + unit(hash2(seed + 1, e));
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_gen_gspd(int32_t nblocks, uint64_t seed, int32_t *Ap, int32_t *Ai,
                                                  double *Ax) {
    __shared__ double R[BS][BS + 1];
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < BS * BS; t += blockDim.x) {
        const int r = t / BS, k = t % BS;
        const uint64_t c = ((uint64_t)b * BS + (uint64_t)r) * BS + (uint64_t)k;
        R[r][k] = -1.0 + 2.0 * unit(hash2(seed, c));
    }
    __syncthreads();
    for (int t = threadIdx.x; t < BS * BS; t += blockDim.x) {
        const int c = t / BS, r = t % BS;  // column c of the block, row r
        double acc = 0.0;
        for (int k = 0; k < BS; k++) {
            double prod = R[r][k] * R[c][k];
            acc = acc + prod;
        }
        double v = acc / (double)BS;
        if (r == c) v = v + (double)BS;
        const int64_t q = ((int64_t)b * BS + c) * BS + r;
        Ai[q] = b * BS + r;
        Ax[q] = v;
    }
    for (int c = threadIdx.x; c < BS; c += blockDim.x) Ap[(int64_t)b * BS + c] = (int32_t)(((int64_t)b * BS + c) * BS);
    if (b == nblocks - 1 && threadIdx.x == 0) Ap[(int64_t)nblocks * BS] = (int32_t)((int64_t)nblocks * BS * BS);
}

__global__ __launch_bounds__(256) void k_gen_vec(int64_t len, uint64_t seed, double lo, double hi, double *v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double span = hi - lo;
    double t = span * unit(hash2(seed, (uint64_t)i));
    v[i] = lo + t;
}

__global__ __launch_bounds__(256) void k_gen_rhs(int32_t n, int32_t nrhs, int32_t col0, double *B) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * nrhs) return;
    const int64_t i = t / nrhs, r = t % nrhs;
    double q = (double)(i + col0 + r) / (double)n;
    B[t] = 1.0 + q;
}

#pragma clang fp contract(fast)

}  // namespace csx

using namespace csx;

extern "C" int csx_gen_grand(int32_t n, int32_t per_col, uint64_t seed, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (n <= 0 || per_col <= 0 || per_col > n || (int64_t)n * per_col > 2147483647ll || !out) return CSX_EINVAL;
    const int32_t nnz = n * per_col;
    CSX_TRY(csx_csc_alloc(n, n, nnz, 1, out));
    Csc *A = csc(*out);
    int64_t blocks = ((int64_t)nnz + 1 + 255) / 256;
    hipLaunchKernelGGL(k_gen_grand, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, n, per_col, seed, A->p, A->i,
                       A->x);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

extern "C" int csx_gen_grand_uniform(int32_t n, int32_t per_col, uint64_t seed, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (n <= 0 || per_col <= 0 || per_col > 64 || 2 * (int64_t)per_col > n || (int64_t)n * per_col > 2147483647ll || !out)
        return CSX_EINVAL;
    const int32_t nnz = n * per_col;
    CSX_TRY(csx_csc_alloc(n, n, nnz, 1, out));
    Csc *A = csc(*out);
    int64_t blocks = ((int64_t)n + 1 + 3) / 4;  // 4 waves (columns) per workgroup; column n only writes Ap[n]
    hipLaunchKernelGGL(k_gen_grand_uniform, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, n, per_col, seed, A->p,
                       A->i, A->x);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

extern "C" int csx_gen_gspd(int32_t nblocks, int32_t bs, uint64_t seed, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (nblocks <= 0 || !out || (bs != 4 && bs != 8 && bs != 16 && bs != 32 && bs != 64)) return CSX_EINVAL;
    const int64_t n = (int64_t)nblocks * bs, nnz = n * bs;
    if (nnz > 2147483647ll) return CSX_EINVAL;
    CSX_TRY(csx_csc_alloc((int32_t)n, (int32_t)n, (int32_t)nnz, 1, out));
    Csc *A = csc(*out);
    hipStream_t s = ctx().stream;
#define CSX_GSPD(BS) \
    hipLaunchKernelGGL(k_gen_gspd<BS>, dim3((unsigned)nblocks), dim3(256), 0, s, nblocks, seed, A->p, A->i, A->x)
    switch (bs) {
        case 4: CSX_GSPD(4); break;
        case 8: CSX_GSPD(8); break;
        case 16: CSX_GSPD(16); break;
        case 32: CSX_GSPD(32); break;
        default: CSX_GSPD(64); break;
    }
#undef CSX_GSPD
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

extern "C" int csx_gen_vec(int64_t len, uint64_t seed, double lo, double hi, csx_handle_t *out) {
    CSX_TRY(csx_vec_alloc(len, out));
    if (len == 0) return CSX_OK;
    Vec *v = vec(*out);
    int64_t blocks = (len + 255) / 256;
    hipLaunchKernelGGL(k_gen_vec, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, len, seed, lo, hi, (double *)v->d);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

extern "C" int csx_gen_rhs(int32_t n, int32_t nrhs, int32_t col0, csx_handle_t *out) {
    if (n <= 0 || nrhs <= 0) return CSX_EINVAL;
    CSX_TRY(csx_vec_alloc((int64_t)n * nrhs, out));
    Vec *v = vec(*out);
    int64_t blocks = ((int64_t)n * nrhs + 255) / 256;
    hipLaunchKernelGGL(k_gen_rhs, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, n, nrhs, col0, (double *)v->d);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}
