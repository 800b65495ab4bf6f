// cs_gaxpy (csparse.py:1199-1213), LDS-tiled plan for matrices whose rows share
// no columns (uniformly random structure, the "G-rand" benchmark input).
//
// Why: with y and x of 40 MB each (n = 5e6) every one of the 3.2e8 entries of A
// makes one random 8-byte access to a vector that fits neither LDS nor an XCD's
// 4 MiB L2.  Gathering x row by row fetches a 128-byte line per entry; scattering
// into y needs memory-side atomics.  Both run far below the HBM rate.
//
// Plan: regroup the entries once (stable radix sort, csx_sort.hip) into tiles
//      tile(s, b) = { A(i, j) : j in column slab s, i in row block b }
// kept in column order inside a tile.  A row block is 2^RB rows, so its slice of
// y lives in LDS (128 KiB at RB = 14) and takes the random accumulation as
// ds_add_f64; a column slab is <= 2.5 MB of x, so it stays in the 4 MiB L2 of
// the XCD that works on it, and because the tile is column-sorted neighbouring
// lanes hit the same 128-byte lines of x.  The entry stream itself (4-byte
// packed (col,row) key + 8-byte value = the same 12 bytes/entry as CSC) is read
// once, coalesced.  Each (slab, row block) tile then writes its y slice to a
// per-slab partial buffer, and a second kernel adds the slabs' partials to y in
// slab order.
//
// XCD awareness: slab s is served by work queue s % 8; a workgroup reads its
// XCC id and drains "its" queue first, then helps the others, so placement only
// affects speed, never the result.  Counters are zeroed before every launch.
//
// HBM bytes per call: 12 nnz (entries) + 2 * 8 * m * nslab (partials, written
// then read) + 16 m (y) + 8 n (x)  -- the partials are the price of the plan.
#include <cstdlib>

#include "csx_internal.h"

namespace csx {

constexpr int TL_THREADS = 1024;
constexpr int TL_QUEUES = 8;

__device__ __forceinline__ unsigned xcc_id() {
    // s_getreg_b32 HW_REG_XCC_ID (id 20), bits [3:0]; only used as a queue preference
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
}

__global__ __launch_bounds__(256) void k_tile_keys(int64_t nnz, const int32_t *__restrict__ Ai,
                                                   const int32_t *__restrict__ col, int rb_bits, int32_t slab_cols,
                                                   int32_t nrb, uint32_t *__restrict__ tile_id,
                                                   uint32_t *__restrict__ packed) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t i = (uint32_t)Ai[p], j = (uint32_t)col[p];
    const uint32_t s = j / (uint32_t)slab_cols, lc = j - s * (uint32_t)slab_cols;
    const uint32_t b = i >> rb_bits, lr = i & ((1u << rb_bits) - 1u);
    tile_id[p] = s * (uint32_t)nrb + b;
    packed[p] = (lc << rb_bits) | lr;
}

template <int RB_BITS>
__global__ __launch_bounds__(TL_THREADS) void k_gaxpy_tiled(const int32_t *__restrict__ tile_ptr,
                                                            const uint32_t *__restrict__ tile_key,
                                                            const double *__restrict__ tile_val,
                                                            const double *__restrict__ x, double *__restrict__ partial,
                                                            int32_t *queue, int32_t nrb, int32_t nslab,
                                                            int32_t slab_cols, int64_t mpad) {
    constexpr int RB = 1 << RB_BITS;
    // all LDS in the dynamic region (16-byte aligned base): RB doubles + one work-item word
    extern __shared__ __attribute__((aligned(16))) double ytile[];
    int &s_item = *reinterpret_cast<int *>(ytile + RB);
    const unsigned home = xcc_id();
    for (int hop = 0; hop < TL_QUEUES; hop++) {
        const int qid = (int)((home + hop) & (TL_QUEUES - 1));
        // queue qid serves slabs qid, qid + 8, ...: all row blocks of one slab before the next
        const int nslab_q = (nslab - qid + TL_QUEUES - 1) / TL_QUEUES;
        const int items = nslab_q > 0 ? nslab_q * nrb : 0;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_item = items > 0 ? atomicAdd(&queue[qid], 1) : items;
            __syncthreads();
            const int t = s_item;
            if (t >= items) break;
            const int slab = qid + TL_QUEUES * (t / nrb);
            const int rb = t % nrb;
            for (int k = threadIdx.x; k < RB; k += TL_THREADS) ytile[k] = 0.0;
            __syncthreads();
            const int tile = slab * nrb + rb;
            const int32_t b = tile_ptr[tile], e = tile_ptr[tile + 1];
            const double *xs = x + (int64_t)slab * slab_cols;
            // two entries per lane per step; b may be odd, so peel to an even position
            int32_t q = b + 2 * (int32_t)threadIdx.x;
            const int32_t b2 = (b + 1) & ~1;
            if (b2 != b && b < e) {
                if (threadIdx.x == 0) {
                    const uint32_t kk = tile_key[b];
                    unsafeAtomicAdd(&ytile[kk & (RB - 1)], tile_val[b] * xs[kk >> RB_BITS]);
                }
                q = b2 + 2 * (int32_t)threadIdx.x;
            }
            for (; q + 1 < e; q += 2 * TL_THREADS) {
                const uint2 kk = *reinterpret_cast<const uint2 *>(tile_key + q);
                const double2 vv = *reinterpret_cast<const double2 *>(tile_val + q);
                const double x0 = xs[kk.x >> RB_BITS], x1 = xs[kk.y >> RB_BITS];
                unsafeAtomicAdd(&ytile[kk.x & (RB - 1)], vv.x * x0);
                unsafeAtomicAdd(&ytile[kk.y & (RB - 1)], vv.y * x1);
            }
            if (q < e) {  // odd tail
                const uint32_t kk = tile_key[q];
                unsafeAtomicAdd(&ytile[kk & (RB - 1)], tile_val[q] * xs[kk >> RB_BITS]);
            }
            __syncthreads();
            double2 *dst = reinterpret_cast<double2 *>(partial + (int64_t)slab * mpad + (int64_t)rb * RB);
            const double2 *src = reinterpret_cast<const double2 *>(ytile);
            for (int k = threadIdx.x; k < RB / 2; k += TL_THREADS) dst[k] = src[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_reduce_partials(int64_t m, int32_t nslab, int64_t mpad,
                                                         const double *__restrict__ partial, double *__restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double acc = y[i];
    for (int s = 0; s < nslab; s++) acc += partial[(int64_t)s * mpad + i];
    y[i] = acc;
}

int gaxpy_tiled_prepare(Csc *A) {
    if (A->tiled) return CSX_OK;
    if (!A->x) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    int rb_bits = 14;
    if (const char *e = std::getenv("CSX_TILED_RB_BITS")) rb_bits = std::atoi(e) == 13 ? 13 : 14;
    double slab_mb = 2.5;
    if (const char *e = std::getenv("CSX_TILED_SLAB_MB")) slab_mb = std::atof(e) > 0.1 ? std::atof(e) : slab_mb;
    const int64_t max_cols_key = 1ll << (32 - rb_bits);
    int64_t max_cols = (int64_t)(slab_mb * 1024 * 1024 / 8);
    if (max_cols > max_cols_key) max_cols = max_cols_key;
    int nslab = TL_QUEUES;
    while (((int64_t)A->n + nslab - 1) / nslab > max_cols) nslab += TL_QUEUES;
    int32_t slab_cols = (int32_t)(((int64_t)A->n + nslab - 1) / nslab);
    if (slab_cols < 1) slab_cols = 1;
    nslab = (int)(((int64_t)A->n + slab_cols - 1) / slab_cols);
    if (nslab < 1) nslab = 1;
    const int32_t rb = 1 << rb_bits;
    const int32_t nrb = (int32_t)(((int64_t)A->m + rb - 1) / rb);

    TiledPlan *t = new TiledPlan();
    t->rb_bits = rb_bits;
    t->row_block = rb;
    t->nrb = nrb;
    t->nslab = nslab;
    t->slab_cols = slab_cols;
    t->ngroup = nslab;
    const int64_t ntiles = (int64_t)nslab * nrb;
    const int64_t mpad = (int64_t)nrb * rb;
    int32_t *col = nullptr;
    uint32_t *tid = nullptr, *packed = nullptr, *stid = nullptr;
    int st = dalloc(&t->tile_ptr, (size_t)ntiles + 1);
    if (st == CSX_OK) st = dalloc(&t->tile_key, (size_t)A->nnz + 2);
    if (st == CSX_OK) st = dalloc(&t->tile_val, (size_t)A->nnz + 2);
    if (st == CSX_OK) st = dalloc(&t->partial, (size_t)(mpad * nslab));
    if (st == CSX_OK) st = dalloc(&t->queue, (size_t)TL_QUEUES);
    if (st == CSX_OK) st = dalloc(&col, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&tid, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&packed, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&stid, (size_t)A->nnz);
    if (st == CSX_OK) st = expand_columns(A->p, A->n, A->nnz, col);
    if (st == CSX_OK && A->nnz > 0) {
        int64_t blocks = ((int64_t)A->nnz + 255) / 256;
        hipLaunchKernelGGL(k_tile_keys, dim3((unsigned)blocks), dim3(256), 0, s, (int64_t)A->nnz, A->i, col, rb_bits,
                           slab_cols, nrb, tid, packed);
        if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
    }
    if (st == CSX_OK)
        st = stable_sort_by_key(tid, packed, A->x, A->nnz, (uint32_t)ntiles, stid, t->tile_key, t->tile_val);
    if (st == CSX_OK) st = boundaries_from_sorted(stid, A->nnz, (int32_t)ntiles, t->tile_ptr);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    dfree(col);
    dfree(tid);
    dfree(packed);
    dfree(stid);
    if (st != CSX_OK) {
        free_tiled(t);
        return st;
    }
    A->tiled = t;
    return CSX_OK;
}

int gaxpy_tiled_run(const Csc *A, const double *x, double *y) {
    const TiledPlan *t = A->tiled;
    hipStream_t s = ctx().stream;
    CSX_HIP(hipMemsetAsync(t->queue, 0, TL_QUEUES * sizeof(int32_t), s));
    const int64_t mpad = (int64_t)t->nrb * t->row_block;
    const size_t lds = (size_t)t->row_block * sizeof(double) + 16;
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KiB of dynamic LDS has to be requested explicitly
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gaxpy_tiled<13>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (1 << 13) * 8 + 16));
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gaxpy_tiled<14>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (1 << 14) * 8 + 16));
        attr_set = true;
    }
    const int wg_per_cu = t->rb_bits == 13 ? 2 : 1;
    const unsigned grid = (unsigned)(ctx().cus * wg_per_cu);
    if (t->rb_bits == 13)
        hipLaunchKernelGGL(k_gaxpy_tiled<13>, dim3(grid), dim3(TL_THREADS), lds, s, t->tile_ptr, t->tile_key,
                           t->tile_val, x, t->partial, t->queue, t->nrb, t->nslab, t->slab_cols, mpad);
    else
        hipLaunchKernelGGL(k_gaxpy_tiled<14>, dim3(grid), dim3(TL_THREADS), lds, s, t->tile_ptr, t->tile_key,
                           t->tile_val, x, t->partial, t->queue, t->nrb, t->nslab, t->slab_cols, mpad);
    CSX_LAUNCH_CHECK();
    int64_t blocks = ((int64_t)A->m + 255) / 256;
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)blocks), dim3(256), 0, s, (int64_t)A->m, t->nslab, mpad,
                       t->partial, y);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx
