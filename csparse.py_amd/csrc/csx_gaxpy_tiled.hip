// cs_gaxpy (csparse.py:1199-1213), LDS-resident plan for matrices whose rows share
// no columns (uniformly random structure, the "G-rand" benchmark input).
//
// Why: with y and x of 40 MB each (n = 5e6) every one of the 3.2e8 entries of A
// makes one random 8-byte access to a vector that fits neither LDS nor an XCD's
// 4 MiB L2.  Gathering x row by row fetches a 128-byte line per entry; scattering
// into y needs memory-side atomics.  Both run far below the HBM rate.
//
// Plan: regroup the entries once (stable radix sort, csx_sort.hip) into tiles
//      tile(b, s) = { A(i, j) : i in row block b, j in column slab s },
// row-block-major, column order kept inside a tile.
//   * One workgroup owns one row block for the whole kernel: its slice of y
//     (m / 256 rows, up to 156 KiB) is loaded into LDS once, takes every random
//     accumulation as ds_add_f64, and is written back once.  No partial sums in
//     memory, no second kernel, no inter-workgroup communication.
//   * All workgroups walk the column slabs in the same order and at the same
//     pace (equal work per tile for a random matrix), so the slab of x in use
//     (<= 1 MiB) is shared through each XCD's L2.
//   * Entries are stored in groups of 256 (tiles padded to a multiple of 256),
//     INTERLEAVED so that the 16-byte key load of lane l returns entries
//     l, l+64, l+128, l+192 of the group: the k-th gather instruction of a wave
//     then covers 64 CONSECUTIVE entries of the column-sorted tile, i.e. ~16
//     cache lines of x instead of 64.  Measured: the texture addresser handles
//     about one distinct line per clock, and with 4 consecutive entries per lane
//     the gathers alone cost 0.6 ms of a 1.3 ms kernel (profiles/ablation_r01.md).
//   * The entry stream (4-byte packed (col,row) key + 8-byte value = the same 12
//     bytes per entry as CSC) is read once, coalesced, non-temporal, one group
//     ahead of the gathers.
//
// HBM bytes per call ~ 12 nnz (+0.4 % padding) + 16 m + 8 n * (#XCDs that read x).
#include <cstdlib>

#include "csx_internal.h"

namespace csx {

constexpr int TL_THREADS = 1024;
constexpr int TL_WAVES = TL_THREADS / 64;
constexpr int TL_GROUP = 256;                   // entries per group = 64 lanes x 4
constexpr int TL_LDS_BYTES = 160 * 1024 - 256;  // leave a little headroom below the CU's 160 KiB
constexpr int TL_LDS_ROWS = TL_LDS_BYTES / 8 - 2;  // y rows per tile (one slot is the padding dummy)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_tile_keys(int64_t nnz, const int32_t *__restrict__ Ai,
                                                   const int32_t *__restrict__ col, int rb_bits, int32_t row_block,
                                                   int32_t slab_cols, int32_t nslab, uint32_t *__restrict__ tile_id,
                                                   uint32_t *__restrict__ packed) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t i = (uint32_t)Ai[p], j = (uint32_t)col[p];
    const uint32_t s = j / (uint32_t)slab_cols, lc = j - s * (uint32_t)slab_cols;
    const uint32_t b = i / (uint32_t)row_block, lr = i - b * (uint32_t)row_block;
    tile_id[p] = b * (uint32_t)nslab + s;
    packed[p] = (lc << rb_bits) | lr;
}

__global__ __launch_bounds__(256) void k_padded_lengths(int64_t ntiles, const int32_t *__restrict__ sptr,
                                                        int32_t *__restrict__ ngroups) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) ngroups[t] = (sptr[t + 1] - sptr[t] + TL_GROUP - 1) / TL_GROUP;
}

// group_info[g] = (slab << 9) | entries in the group (1..256)
__global__ __launch_bounds__(256) void k_group_info(int64_t ntiles, int32_t nslab, const int32_t *__restrict__ sptr,
                                                    const int32_t *__restrict__ gptr, uint32_t *__restrict__ info) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const int32_t len = sptr[t + 1] - sptr[t];
    const uint32_t slab = (uint32_t)(t % nslab);
    int32_t g = gptr[t];
    for (int32_t left = len; left > 0; left -= TL_GROUP, g++) info[g] = (slab << 9) | (uint32_t)min(left, TL_GROUP);
}

// sorted position q (tile t, rank e inside the tile) -> interleaved slot
__global__ __launch_bounds__(256) void k_interleave(int64_t nnz, const uint32_t *__restrict__ stid,
                                                    const int32_t *__restrict__ sptr, const int32_t *__restrict__ gptr,
                                                    const uint32_t *__restrict__ skey, const double *__restrict__ sval,
                                                    uint32_t *__restrict__ okey, double *__restrict__ oval) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const uint32_t t = stid[q];
    const int32_t e = (int32_t)(q - sptr[t]);
    const int32_t g = e / TL_GROUP, w = e % TL_GROUP;
    // entry w of a group goes to lane l = w & 63, component k = w >> 6.  Keys: lane l's 16-byte
    // word (slots 4l..4l+3).  Values: two planes of 128 doubles, plane k >> 1, lane l's 16-byte
    // word (slots 2l, 2l+1) -- so that EACH value load instruction of a wave covers one contiguous
    // 1 KiB (whole 128-byte lines).  With values at slots 4l..4l+3 each instruction touched only
    // half of every line and issued twice as many (64-byte) requests.
    const int64_t gbase = ((int64_t)gptr[t] + g) * TL_GROUP;
    const int l = w & 63, k = w >> 6;
    okey[gbase + l * 4 + k] = skey[q];
    oval[gbase + (k >> 1) * 128 + l * 2 + (k & 1)] = sval[q];
}

__global__ __launch_bounds__(256) void k_fill_keys(uint32_t *p, int64_t n, uint32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

struct GroupRegs {
    u32x4 kk;
    f64x2 v0, v1;
    uint32_t info;
};

template <bool NT>
__device__ __forceinline__ GroupRegs load_group_t(const uint32_t *__restrict__ key, const double *__restrict__ val,
                                                  const uint32_t *__restrict__ info, int32_t g, int lane) {
    GroupRegs r;
    const int64_t gb = (int64_t)g * TL_GROUP;
    if (NT) {
        r.kk = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(key + gb + 4 * lane));
        r.v0 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + gb + 2 * lane));
        r.v1 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + gb + 128 + 2 * lane));
    } else {
        r.kk = *reinterpret_cast<const u32x4 *>(key + gb + 4 * lane);
        r.v0 = *reinterpret_cast<const f64x2 *>(val + gb + 2 * lane);
        r.v1 = *reinterpret_cast<const f64x2 *>(val + gb + 128 + 2 * lane);
    }
    r.info = info[g];
    return r;
}

// how x is gathered: 0 plain (L1-allocating), 1 sc1 (agent-scope relaxed: bypasses L1), 2 non-temporal
template <int HOW>
__device__ __forceinline__ double load_x(const double *p) {
    if (HOW == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (HOW == 2) return __builtin_nontemporal_load(p);
    return *p;
}

// VARIANT bits (timing experiments only; results are wrong unless VARIANT == 0):
//   1 = skip the x gather, 2 = full kernel with plain instead of nt stream loads, 3 = stream only,
//   4 = gather from a 2 KB footprint (L1 hits),
//   5 = gather from a 256 KB footprint (L2 hits, L1 misses)
//   6 = full kernel, x gathered with L1-bypassing sc1 loads (computes y)
//   7 = real x gathers, but the entry stream re-reads the first 32 groups of the row block (L2-resident)
template <int VARIANT>
__global__ __launch_bounds__(TL_THREADS) void k_gaxpy_tiled(const int32_t *__restrict__ rb_gptr,
                                                            const uint32_t *__restrict__ group_info,
                                                            const uint32_t *__restrict__ tile_key,
                                                            const double *__restrict__ tile_val,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            int32_t m, int32_t nrb, int32_t row_block,
                                                            int32_t slab_cols, int rb_bits) {
    extern __shared__ __attribute__((aligned(16))) double ytile[];  // row_block doubles
    constexpr bool GATHER = !(VARIANT & 1) || VARIANT >= 5, ATOMIC = VARIANT != 3;
    constexpr bool STREAM_NT = VARIANT != 2;  // variant 2: full kernel with plain (L1-allocating) stream loads
#define load_group load_group_t<STREAM_NT>
    constexpr uint32_t CMASK = VARIANT == 5 ? 32767u : (VARIANT == 4 ? 255u : 0xffffffffu);
    constexpr int XLOAD = VARIANT == 6 ? 1 : 0;
    constexpr bool RING = VARIANT == 7;
    const uint32_t rmask = (1u << rb_bits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sink = 0.0;
    for (int32_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row0 = (int64_t)rb * row_block;
        const int32_t rows = (int32_t)((row0 + row_block <= m) ? row_block : (m - row0));
        for (int k = threadIdx.x; k < rows; k += TL_THREADS) ytile[k] = y[row0 + k];
        __syncthreads();
        const int32_t gend = rb_gptr[rb + 1];
        const int32_t g0 = rb_gptr[rb];
        // Each wave walks its groups two at a time (g and g + TL_WAVES) with two register sets
        // used alternately (no copies): while set A is gathered and accumulated, set B's entries
        // are in flight.  Prefetch indices are clamped to the last group instead of predicated,
        // and padding entries point at a dummy LDS row, so the loop body is branch-free.
#define CSX_GIDX(gg) (RING ? g0 + (((gg) - g0) & 31) : (gg))
#define CSX_LOADPAIR(ra, rb2, gg)                                                                          \
    {                                                                                                      \
        const int32_t ga_ = (gg) < gend ? (gg) : gend - 1;                                                 \
        const int32_t gb_ = (gg) + TL_WAVES < gend ? (gg) + TL_WAVES : gend - 1;                           \
        ra = load_group(tile_key, tile_val, group_info, CSX_GIDX(ga_), lane);                              \
        rb2 = load_group(tile_key, tile_val, group_info, CSX_GIDX(gb_), lane);                             \
    }
#define CSX_GATHER(c, xa, xb, xc, xd)                                    \
    {                                                                    \
        const double *xs_ = x + (int64_t)(c.info >> 9) * slab_cols;      \
        if (GATHER) {                                                    \
            xa = load_x<XLOAD>(xs_ + ((c.kk.x >> rb_bits) & CMASK));     \
            xb = load_x<XLOAD>(xs_ + ((c.kk.y >> rb_bits) & CMASK));     \
            xc = load_x<XLOAD>(xs_ + ((c.kk.z >> rb_bits) & CMASK));     \
            xd = load_x<XLOAD>(xs_ + ((c.kk.w >> rb_bits) & CMASK));     \
        }                                                                \
    }
#define CSX_ACCUM(c, xa, xb, xc, xd)                                                        \
    {                                                                                       \
        if (ATOMIC) {                                                                       \
            unsafeAtomicAdd(&ytile[c.kk.x & rmask], c.v0.x * xa);                           \
            unsafeAtomicAdd(&ytile[c.kk.y & rmask], c.v0.y * xb);                           \
            unsafeAtomicAdd(&ytile[c.kk.z & rmask], c.v1.x * xc);                           \
            unsafeAtomicAdd(&ytile[c.kk.w & rmask], c.v1.y * xd);                           \
        } else {                                                                            \
            sink += c.v0.x * xa + c.v0.y * xb + c.v1.x * xc + c.v1.y * xd +                 \
                    (double)((c.kk.x ^ c.kk.y ^ c.kk.z ^ c.kk.w) & 1u);                     \
        }                                                                                   \
    }
#define CSX_STEP(ca, cb, na, nb)                                                                          \
    {                                                                                                     \
        const bool two_ = g + TL_WAVES < gend;                                                            \
        double a0 = 1.0, a1 = 1.0, a2 = 1.0, a3 = 1.0, b0 = 1.0, b1 = 1.0, b2 = 1.0, b3 = 1.0;            \
        CSX_GATHER(ca, a0, a1, a2, a3)                                                                    \
        if (two_) CSX_GATHER(cb, b0, b1, b2, b3)                                                          \
        CSX_LOADPAIR(na, nb, g + 2 * TL_WAVES) /* behind the gathers */                                   \
        CSX_ACCUM(ca, a0, a1, a2, a3)                                                                     \
        if (two_) CSX_ACCUM(cb, b0, b1, b2, b3)                                                           \
        g += 2 * TL_WAVES;                                                                                \
    }
        int32_t g = g0 + wave;
        if (g < gend) {
            GroupRegs pa, pb, qa, qb;
            CSX_LOADPAIR(pa, pb, g)
            for (;;) {
                CSX_STEP(pa, pb, qa, qb)
                if (g >= gend) break;
                CSX_STEP(qa, qb, pa, pb)
                if (g >= gend) break;
            }
        }
#undef CSX_STEP
#undef CSX_GATHER
#undef CSX_ACCUM
#undef CSX_LOADPAIR
#undef CSX_GIDX
#undef load_group
        __syncthreads();
        if (!ATOMIC && sink == 12345.678) ytile[0] = sink;  // keep the ablated arithmetic alive
        for (int k = threadIdx.x; k < rows; k += TL_THREADS) y[row0 + k] = ytile[k];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rb_group_ptr(int32_t nrb, int32_t nslab, const int32_t *__restrict__ gptr,
                                                      int32_t *__restrict__ rb_gptr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nrb) rb_gptr[b] = gptr[(int64_t)b * nslab];
}

int gaxpy_tiled_prepare(Csc *A) {
    if (A->tiled) return CSX_OK;
    if (!A->x) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    // one row block per workgroup, one workgroup per CU; more rounds only if a block would not fit LDS
    int wg_per_cu = 1;
    if (const char *e = std::getenv("CSX_TILED_WG_PER_CU")) wg_per_cu = std::atoi(e) == 2 ? 2 : 1;
    const int32_t nwg = (ctx().cus > 0 ? ctx().cus : 256) * wg_per_cu;
    const int32_t cap = TL_LDS_ROWS;
    int32_t rounds = 1;
    int32_t row_block;
    for (;;) {
        const int64_t nrb_try = (int64_t)nwg * rounds;
        row_block = (int32_t)(((int64_t)A->m + nrb_try - 1) / nrb_try);
        if (row_block <= cap) break;
        rounds++;
    }
    if (row_block < 1) row_block = 1;
    const int32_t nrb = (int32_t)(((int64_t)A->m + row_block - 1) / row_block);
    int rb_bits = 1;
    while ((1 << rb_bits) <= row_block) rb_bits++;  // local row index row_block itself = dummy slot for padding
    double slab_kb = 1024.0;
    if (const char *e = std::getenv("CSX_TILED_SLAB_KB")) slab_kb = std::atof(e) >= 8.0 ? std::atof(e) : slab_kb;
    int64_t slab_cols = (int64_t)(slab_kb * 1024 / 8);
    const int64_t max_cols_key = 1ll << (32 - rb_bits);
    if (slab_cols > max_cols_key) slab_cols = max_cols_key;
    if (slab_cols > A->n) slab_cols = A->n > 0 ? A->n : 1;
    int32_t nslab = (int32_t)(((int64_t)A->n + slab_cols - 1) / slab_cols);
    if (nslab < 1) nslab = 1;
    const int64_t ntiles = (int64_t)nrb * nslab;
    if (ntiles > 0x3fffffffll || nslab >= (1 << 22)) return CSX_EINVAL;

    TiledPlan *t = new TiledPlan();
    t->rb_bits = rb_bits;
    t->row_block = row_block;
    t->nrb = nrb;
    t->nslab = nslab;
    t->slab_cols = (int32_t)slab_cols;
    int32_t *col = nullptr, *sptr = nullptr, *gptr = nullptr;
    uint32_t *tid = nullptr, *packed = nullptr, *stid = nullptr, *skey = nullptr;
    double *sval = nullptr;
    int64_t ngroups = 0;
    int st = dalloc(&col, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&tid, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&packed, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&stid, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&skey, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&sval, (size_t)A->nnz);
    if (st == CSX_OK) st = dalloc(&sptr, (size_t)ntiles + 1);
    if (st == CSX_OK) st = dalloc(&gptr, (size_t)ntiles + 1);
    if (st == CSX_OK) st = dalloc(&t->tile_ptr, (size_t)nrb + 1);
    if (st == CSX_OK) st = expand_columns(A->p, A->n, A->nnz, col);
    if (st == CSX_OK && A->nnz > 0) {
        int64_t blocks = ((int64_t)A->nnz + 255) / 256;
        hipLaunchKernelGGL(k_tile_keys, dim3((unsigned)blocks), dim3(256), 0, s, (int64_t)A->nnz, A->i, col, rb_bits,
                           row_block, t->slab_cols, t->nslab, tid, packed);
        if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
    }
    if (st == CSX_OK) st = stable_sort_by_key(tid, packed, A->x, A->nnz, (uint32_t)ntiles, stid, skey, sval);
    if (st == CSX_OK) st = boundaries_from_sorted(stid, A->nnz, (int32_t)ntiles, sptr);
    const unsigned tb = (unsigned)((ntiles + 256) / 256);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_padded_lengths, dim3(tb), dim3(256), 0, s, ntiles, sptr, gptr);
        st = scan_exclusive_i32(gptr, gptr, ntiles, &ngroups);
    }
    if (st == CSX_OK) st = dalloc(&t->tile_len, (size_t)ngroups);  // group_info
    if (st == CSX_OK) st = dalloc(&t->tile_key, (size_t)ngroups * TL_GROUP);
    if (st == CSX_OK) st = dalloc(&t->tile_val, (size_t)ngroups * TL_GROUP);
    if (st == CSX_OK && ngroups > 0) {
        // padding slots: column 0 of the slab, dummy row, value 0 -> adds 0 * x into an unused LDS slot
        hipLaunchKernelGGL(k_fill_keys, dim3(2048), dim3(256), 0, s, t->tile_key, (int64_t)ngroups * TL_GROUP,
                           (uint32_t)row_block);
        (void)hipMemsetAsync(t->tile_val, 0, (size_t)ngroups * TL_GROUP * sizeof(double), s);
        hipLaunchKernelGGL(k_group_info, dim3(tb), dim3(256), 0, s, ntiles, nslab, sptr, gptr, (uint32_t *)t->tile_len);
        hipLaunchKernelGGL(k_interleave, dim3((unsigned)(((int64_t)A->nnz + 255) / 256)), dim3(256), 0, s,
                           (int64_t)A->nnz, stid, sptr, gptr, skey, sval, t->tile_key, t->tile_val);
    }
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_rb_group_ptr, dim3((unsigned)((nrb + 256) / 256)), dim3(256), 0, s, nrb, nslab, gptr,
                           t->tile_ptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    }
    dfree(col);
    dfree(tid);
    dfree(packed);
    dfree(stid);
    dfree(skey);
    dfree(sval);
    dfree(sptr);
    dfree(gptr);
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
    const size_t lds = (((size_t)(t->row_block + 1) * sizeof(double)) + 15) & ~(size_t)15;  // + dummy row
    int variant = 0;
    if (const char *e = std::getenv("CSX_TILED_VARIANT")) variant = std::atoi(e) & 7;
    if (std::getenv("CSX_TILED_XLOAD")) variant = 5 + std::atoi(std::getenv("CSX_TILED_XLOAD"));  // 1 -> sc1, 2 -> nt
    int wg_per_cu = 1;
    if (const char *e = std::getenv("CSX_TILED_WG_PER_CU")) wg_per_cu = std::atoi(e) == 2 ? 2 : 1;
    const int32_t nwg = (ctx().cus > 0 ? ctx().cus : 256) * wg_per_cu;
    const unsigned grid = (unsigned)(t->nrb < nwg ? t->nrb : nwg);
#define CSX_TILED_LAUNCH(V)                                                                                          \
    case V: {                                                                                                        \
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gaxpy_tiled<V>),                               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES));                      \
        hipLaunchKernelGGL(k_gaxpy_tiled<V>, dim3(grid), dim3(TL_THREADS), lds, s, t->tile_ptr,                      \
                           (const uint32_t *)t->tile_len, t->tile_key, t->tile_val, x, y, A->m, t->nrb, t->row_block, \
                           t->slab_cols, t->rb_bits);                                                                \
    } break;
    switch (variant) {
        CSX_TILED_LAUNCH(0)
        CSX_TILED_LAUNCH(1)
        CSX_TILED_LAUNCH(2)
        CSX_TILED_LAUNCH(3)
        CSX_TILED_LAUNCH(4)
        CSX_TILED_LAUNCH(5)
        CSX_TILED_LAUNCH(6)
        CSX_TILED_LAUNCH(7)
        default: return CSX_EINVAL;
    }
#undef CSX_TILED_LAUNCH
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx
