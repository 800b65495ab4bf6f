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
//   * 3-byte keys.  A gather instruction covers 64 consecutive entries of the column-sorted tile; when every such
//     run spans fewer than 512 columns (always, for a matrix like G-rand: a run spans 256 +- 32) an entry needs its
//     row in the block (15 bits) and its column's offset from the run's first column (9 bits): 3 bytes, the run's
//     base column being four words per group of 256.  The stream is then 11 bytes per entry instead of 12.  A matrix
//     with a wider run anywhere keeps the 4-byte keys.
//
// HBM bytes per call ~ 12 nnz (+0.4 % padding) + 16 m + 8 n * (#XCDs that read x).
#include <cstdlib>

#include "csx_internal.h"

namespace csx {

constexpr int TL_WAVES = 4, TL_NG = 5;          // waves per workgroup, groups per wave and step of the shipped kernel (see k_gaxpy_tiled)
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
    if (okey) okey[gbase + l * 4 + k] = skey[q];
    oval[gbase + (k >> 1) * 128 + l * 2 + (k & 1)] = sval[q];
}

// 3-byte keys: is every run of 64 consecutive entries of a tile narrower than 512 columns?
__global__ __launch_bounds__(256) void k_run_span(int64_t nnz, const uint32_t *__restrict__ stid,
                                                  const int32_t *__restrict__ sptr, const uint32_t *__restrict__ skey,
                                                  int rb_bits, int *too_wide) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const int32_t e = (int32_t)(q - sptr[stid[q]]);
    const int64_t first = q - (e & 63);
    if ((skey[q] >> rb_bits) - (skey[first] >> rb_bits) >= 512u) *too_wide = 1;
}

// key24 of entry w of a group: bytes 12 l + 3 k .. + 2 of the group's 768 (lane l = w & 63, run k = w >> 6), value
// row | (column - first column of the run) << 15; base[4 g + k] = that first column
__global__ __launch_bounds__(256) void k_interleave24(int64_t nnz, const uint32_t *__restrict__ stid,
                                                      const int32_t *__restrict__ sptr, const int32_t *__restrict__ gptr,
                                                      const uint32_t *__restrict__ skey, int rb_bits,
                                                      uint8_t *__restrict__ okey, uint32_t *__restrict__ obase) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const uint32_t t = stid[q];
    const int32_t e = (int32_t)(q - sptr[t]);
    const int32_t g = e / TL_GROUP, w = e % TL_GROUP;
    const int l = w & 63, k = w >> 6;
    const uint32_t rmask = (1u << rb_bits) - 1u;
    const uint32_t col = skey[q] >> rb_bits, col0 = skey[q - l] >> rb_bits;
    const uint32_t v = (skey[q] & rmask) | ((col - col0) << 15);
    const int64_t gg = (int64_t)gptr[t] + g;
    uint8_t *o = okey + gg * (TL_GROUP * 3) + l * 12 + k * 3;
    o[0] = (uint8_t)v;
    o[1] = (uint8_t)(v >> 8);
    o[2] = (uint8_t)(v >> 16);
    if (l == 0) obase[gg * 4 + k] = col;
}

__global__ __launch_bounds__(256) void k_fill_key24(uint8_t *p, int64_t nslots, uint32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nslots; i += stride) {
        p[3 * i] = (uint8_t)v;
        p[3 * i + 1] = (uint8_t)(v >> 8);
        p[3 * i + 2] = (uint8_t)(v >> 16);
    }
}

__global__ __launch_bounds__(256) void k_fill_keys(uint32_t *p, int64_t n, uint32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

struct GroupRegs {
    u32x4 kk;      // 4-byte keys as stored; 3-byte keys unpacked to row | offset << 15
    u32x4 base;    // 3-byte keys: first column of the four runs
    f64x2 v0, v1;
    uint32_t info;
};

template <bool NT, bool K24>
__device__ __forceinline__ GroupRegs load_group_t(const uint32_t *__restrict__ key, const double *__restrict__ val,
                                                  const uint32_t *__restrict__ info, const uint32_t *__restrict__ base,
                                                  int32_t g, int lane) {
    GroupRegs r;
    // (Tried: g = readfirstlane(g), which turns the group's info word and base record into scalar loads -- two of nine
    // vector memory instructions per group gone, same time, 2.4 % more bytes at the memory side.  Not kept.)
    const int64_t gb = (int64_t)g * TL_GROUP;
    if (K24) {
        // `key` is the byte array of 3-byte keys: 12 bytes per lane, 768 per group
        const uint32_t *kp = key + (int64_t)g * (TL_GROUP * 3 / 4) + 3 * lane;
        const uint32_t ka = __builtin_nontemporal_load(kp), kb = __builtin_nontemporal_load(kp + 1),
                       kc = __builtin_nontemporal_load(kp + 2);
        r.kk.x = ka & 0xFFFFFFu;
        r.kk.y = (ka >> 24) | ((kb & 0xFFFFu) << 8);
        r.kk.z = (kb >> 16) | ((kc & 0xFFu) << 16);
        r.kk.w = kc >> 8;
        r.base = *reinterpret_cast<const u32x4 *>(base + 4 * (int64_t)g);
        r.v0 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + gb + 2 * lane));
        r.v1 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + gb + 128 + 2 * lane));
    } else if (NT) {
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

// The kernel.  NW waves per workgroup (one workgroup per CU), NG groups per wave and step.
// Measured on G-rand 5M x 5M (profiles/r02_ablation.md): FEWER waves are faster -- 16 waves 0.97 ms, 12 waves
// 0.83 ms, 8 waves 0.86 ms -- the x lines a wave's gathers bring into the 32 KiB L1 are shared by the
// neighbouring lanes' entries only if they survive until the whole gather instruction has been served, and
// the more waves stream through the same L1 the fewer do.
//
// VARIANT (timing experiments, compiled only with -DCSX_ABLATION; the shipped library holds VARIANT 0 alone).
// Results are wrong for variants 1, 3, 4, 5, 7:
//   1 = skip the x gather, 2 = full kernel with plain instead of nt stream loads, 3 = stream only,
//   4 = gather from a 2 KB footprint (L1 hits),
//   5 = gather from a 256 KB footprint (L2 hits, L1 misses)
//   6 = full kernel, x gathered with L1-bypassing sc1 loads (computes y)
//   7 = real x gathers, but the entry stream re-reads the first 32 groups of the row block (L2-resident)
template <int VARIANT, int NW, int NG, bool K24>
struct TiledStep {
    static constexpr bool GATHER = !(VARIANT & 1) || VARIANT >= 5, ATOMIC = VARIANT != 3;
    static constexpr bool STREAM_NT = VARIANT != 2;
    static constexpr uint32_t CMASK = VARIANT == 5 ? 32767u : (VARIANT == 4 ? 255u : 0xffffffffu);
    static constexpr int XLOAD = VARIANT == 6 ? 1 : 0;
    static constexpr bool RING = VARIANT == 7;

    // issue the entry loads of the NG groups g, g + NW, ... (indices clamped to the row block's last group
    // instead of predicated: the loop body stays branch-free)
    static __device__ __forceinline__ void load_set(GroupRegs (&r)[NG], const uint32_t *__restrict__ key,
                                                    const double *__restrict__ val, const uint32_t *__restrict__ info,
                                                    const uint32_t *__restrict__ base, int32_t g, int32_t g0,
                                                    int32_t gend, int lane) {
#pragma unroll
        for (int j = 0; j < NG; j++) {
            int32_t gg = g + j * NW < gend ? g + j * NW : gend - 1;
            if (RING) gg = g0 + ((gg - g0) & 31);
            r[j] = load_group_t<STREAM_NT, K24>(key, val, info, base, gg, lane);
        }
    }

    // gather x for the NG groups in `cur`, put the next NG groups' entry loads behind the gathers, accumulate
    static __device__ __forceinline__ void step(GroupRegs (&cur)[NG], GroupRegs (&nxt)[NG], double *ytile,
                                                const uint32_t *__restrict__ key, const double *__restrict__ val,
                                                const uint32_t *__restrict__ info, const uint32_t *__restrict__ base,
                                                const double *__restrict__ x, int32_t &g, int32_t g0, int32_t gend,
                                                int lane, int rb_bits, uint32_t rmask, int32_t slab_cols, double &sink) {
        double xv[NG][4];
#pragma unroll
        for (int j = 0; j < NG; j++) {
            xv[j][0] = xv[j][1] = xv[j][2] = xv[j][3] = 1.0;
            if (GATHER && (j == 0 || g + j * NW < gend)) {
                const double *xs = x + (int64_t)(cur[j].info >> 9) * slab_cols;
                if (K24) {
                    xv[j][0] = load_x<XLOAD>(xs + ((cur[j].base.x + (cur[j].kk.x >> 15)) & CMASK));
                    xv[j][1] = load_x<XLOAD>(xs + ((cur[j].base.y + (cur[j].kk.y >> 15)) & CMASK));
                    xv[j][2] = load_x<XLOAD>(xs + ((cur[j].base.z + (cur[j].kk.z >> 15)) & CMASK));
                    xv[j][3] = load_x<XLOAD>(xs + ((cur[j].base.w + (cur[j].kk.w >> 15)) & CMASK));
                } else {
                    xv[j][0] = load_x<XLOAD>(xs + ((cur[j].kk.x >> rb_bits) & CMASK));
                    xv[j][1] = load_x<XLOAD>(xs + ((cur[j].kk.y >> rb_bits) & CMASK));
                    xv[j][2] = load_x<XLOAD>(xs + ((cur[j].kk.z >> rb_bits) & CMASK));
                    xv[j][3] = load_x<XLOAD>(xs + ((cur[j].kk.w >> rb_bits) & CMASK));
                }
            }
        }
        load_set(nxt, key, val, info, base, g + NG * NW, g0, gend, lane);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j == 0 || g + j * NW < gend) {
                if (ATOMIC) {
                    unsafeAtomicAdd(&ytile[cur[j].kk.x & rmask], cur[j].v0.x * xv[j][0]);
                    unsafeAtomicAdd(&ytile[cur[j].kk.y & rmask], cur[j].v0.y * xv[j][1]);
                    unsafeAtomicAdd(&ytile[cur[j].kk.z & rmask], cur[j].v1.x * xv[j][2]);
                    unsafeAtomicAdd(&ytile[cur[j].kk.w & rmask], cur[j].v1.y * xv[j][3]);
                } else {
                    sink += cur[j].v0.x * xv[j][0] + cur[j].v0.y * xv[j][1] + cur[j].v1.x * xv[j][2] +
                            cur[j].v1.y * xv[j][3] +
                            (double)((cur[j].kk.x ^ cur[j].kk.y ^ cur[j].kk.z ^ cur[j].kk.w) & 1u);
                }
            }
        }
        g += NG * NW;
    }
};

template <int VARIANT, int NW, int NG, bool K24>
__global__ __launch_bounds__(64 * NW) void k_gaxpy_tiled(const int32_t *__restrict__ rb_gptr,
                                                         const uint32_t *__restrict__ group_info,
                                                         const uint32_t *__restrict__ tile_key,
                                                         const uint32_t *__restrict__ tile_base,
                                                         const double *__restrict__ tile_val,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         int32_t m, int32_t nrb, int32_t row_block,
                                                         int32_t slab_cols, int rb_bits) {
    extern __shared__ __attribute__((aligned(16))) double ytile[];  // row_block doubles + the padding dummy row
    typedef TiledStep<VARIANT, NW, NG, K24> S;
    const uint32_t rmask = K24 ? 0x7FFFu : (1u << rb_bits) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sink = 0.0;
    for (int32_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row0 = (int64_t)rb * row_block;
        const int32_t rows = (int32_t)((row0 + row_block <= m) ? row_block : (m - row0));
        for (int k = threadIdx.x; k < rows; k += 64 * NW) ytile[k] = y[row0 + k];
        __syncthreads();
        const int32_t gend = rb_gptr[rb + 1];
        const int32_t g0 = rb_gptr[rb];
        // Each wave walks its groups NG at a time (g, g + NW, ...) with two register sets used alternately (no
        // copies): while set A is gathered and accumulated, set B's entries are in flight.  Padding entries
        // point at a dummy LDS row.
        int32_t g = g0 + wave;
        if (g < gend) {
            GroupRegs A[NG], B[NG];
            S::load_set(A, tile_key, tile_val, group_info, tile_base, g, g0, gend, lane);
            for (;;) {
                S::step(A, B, ytile, tile_key, tile_val, group_info, tile_base, x, g, g0, gend, lane, rb_bits, rmask, slab_cols, sink);
                if (g >= gend) break;
                S::step(B, A, ytile, tile_key, tile_val, group_info, tile_base, x, g, g0, gend, lane, rb_bits, rmask, slab_cols, sink);
                if (g >= gend) break;
            }
        }
        __syncthreads();
        if (!S::ATOMIC && sink == 12345.678) ytile[0] = sink;  // keep the ablated arithmetic alive
        for (int k = threadIdx.x; k < rows; k += 64 * NW) y[row0 + k] = ytile[k];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rb_group_ptr(int32_t nrb, int32_t nslab, const int32_t *__restrict__ gptr,
                                                      int32_t *__restrict__ rb_gptr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nrb) rb_gptr[b] = gptr[(int64_t)b * nslab];
}

static int gaxpy_tiled_pick_shape(Csc *A);

int gaxpy_tiled_prepare(Csc *A) {
    if (A->tiled) return CSX_OK;
    if (!A->x) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    // one row block per workgroup, one workgroup per CU; more rounds only if a block would not fit LDS
    const int32_t nwg = ctx().cus > 0 ? ctx().cus : 256;
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
#ifdef CSX_ABLATION
    if (const char *e = std::getenv("CSX_TILED_SLAB_KB")) slab_kb = std::atof(e) >= 8.0 ? std::atof(e) : slab_kb;
#endif
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
    // 3-byte keys when every 64-entry run is narrower than 512 columns (and the row fits 15 bits: always, LDS bounds it)
    bool k24 = false;
    if (st == CSX_OK && ngroups > 0 && rb_bits <= 15 && ctx().opt.gaxpy_keys24) {
        int *flag = nullptr;
        int h = 1;
        st = dalloc(&flag, 1);
        if (st == CSX_OK) {
            (void)hipMemsetAsync(flag, 0, sizeof(int), s);
            hipLaunchKernelGGL(k_run_span, dim3((unsigned)(((int64_t)A->nnz + 255) / 256)), dim3(256), 0, s, (int64_t)A->nnz,
                               stid, sptr, skey, rb_bits, flag);
            if (hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        dfree(flag);
        k24 = st == CSX_OK && h == 0;
    }
    if (st == CSX_OK) st = dalloc(&t->tile_len, (size_t)ngroups);  // group_info
    if (st == CSX_OK && !k24) st = dalloc(&t->tile_key, (size_t)ngroups * TL_GROUP);
    if (st == CSX_OK && k24) st = dalloc(&t->tile_key24, (size_t)ngroups * TL_GROUP * 3 + 16);
    if (st == CSX_OK && k24) st = dalloc(&t->tile_base, (size_t)ngroups * 4);
    if (st == CSX_OK) st = dalloc(&t->tile_val, (size_t)ngroups * TL_GROUP);
    if (st == CSX_OK && ngroups > 0) {
        // padding slots: column 0 of the slab / offset 0 of the run, dummy row, value 0 -> adds 0 * x into an unused LDS slot
        if (k24) {
            hipLaunchKernelGGL(k_fill_key24, dim3(2048), dim3(256), 0, s, t->tile_key24, (int64_t)ngroups * TL_GROUP,
                               (uint32_t)row_block);
            (void)hipMemsetAsync(t->tile_base, 0, (size_t)ngroups * 4 * sizeof(uint32_t), s);
        } else {
            hipLaunchKernelGGL(k_fill_keys, dim3(2048), dim3(256), 0, s, t->tile_key, (int64_t)ngroups * TL_GROUP,
                               (uint32_t)row_block);
        }
        (void)hipMemsetAsync(t->tile_val, 0, (size_t)ngroups * TL_GROUP * sizeof(double), s);
        hipLaunchKernelGGL(k_group_info, dim3(tb), dim3(256), 0, s, ntiles, nslab, sptr, gptr, (uint32_t *)t->tile_len);
        hipLaunchKernelGGL(k_interleave, dim3((unsigned)(((int64_t)A->nnz + 255) / 256)), dim3(256), 0, s,
                           (int64_t)A->nnz, stid, sptr, gptr, skey, sval, t->tile_key, t->tile_val);
        if (k24)
            hipLaunchKernelGGL(k_interleave24, dim3((unsigned)(((int64_t)A->nnz + 255) / 256)), dim3(256), 0, s,
                               (int64_t)A->nnz, stid, sptr, gptr, skey, rb_bits, t->tile_key24, t->tile_base);
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
#ifndef CSX_ABLATION
    // Off by default ("gaxpy.tune_shape"): on every box and run seen 4 x 5 was the fastest or within the noise of the
    // fastest, and a choice made from two short timings is itself noisy (under the profiler it picked 8 x 4, which
    // moves 13 % more bytes).  A plan that is not tuned launches 4 x 5.
    if (ctx().opt.gaxpy_tune_shape) CSX_TRY(gaxpy_tiled_pick_shape(A));
#endif
    return CSX_OK;
}

// Launch shapes offered to the plan: the same kernel body, plan and LDS tile, only waves per workgroup x groups per wave
// and step differ.  4 x 5 and 2 x 10 are within 1 % of each other on the boxes measured (profiles/r02_ablation.md, 1),
// 8 x 4 and 2 x 8 within 5 %, and the whole kernel varies by 8 % from box to box -- so the plan times them where it runs.
constexpr int TL_NSHAPES = 4;
static int gaxpy_tiled_launch(const Csc *A, const double *x, double *y, int shape);

static int gaxpy_tiled_pick_shape(Csc *A) {
    TiledPlan *t = A->tiled;
    if ((int64_t)A->nnz < (int64_t)1 << 24) return CSX_OK;      // small matrices: the default shape, no 10 ms of tuning
    hipStream_t s = ctx().stream;
    DevScope tmp;
    double *x = nullptr, *y = nullptr;
    if (tmp.alloc(&x, (size_t)A->n) != CSX_OK || tmp.alloc(&y, (size_t)A->m) != CSX_OK) return CSX_OK;   // no room: default
    CSX_HIP(hipMemsetAsync(x, 0, (size_t)A->n * sizeof(double), s));   // the gathers go where they always go; values do not matter
    CSX_HIP(hipMemsetAsync(y, 0, (size_t)A->m * sizeof(double), s));
    hipEvent_t e0, e1;
    CSX_HIP(hipEventCreate(&e0));
    CSX_HIP(hipEventCreate(&e1));
    int best = 0;
    int st = CSX_OK;
    for (int round = 0; round < 2 && st == CSX_OK; round++)             // second round: best of two per shape
        for (int sh = 0; sh < TL_NSHAPES && st == CSX_OK; sh++) {
            if (round == 0) st = gaxpy_tiled_launch(A, x, y, sh);       // not timed: first launch of this instantiation
            if (st != CSX_OK) break;
            (void)hipEventRecord(e0, s);
            for (int rep = 0; rep < 3 && st == CSX_OK; rep++) st = gaxpy_tiled_launch(A, x, y, sh);
            (void)hipEventRecord(e1, s);
            if (hipEventSynchronize(e1) != hipSuccess) st = CSX_ERUNTIME;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            if (round == 0 || ms < t->shape_ms[sh]) t->shape_ms[sh] = ms;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (st != CSX_OK) return st;
    for (int sh = 1; sh < TL_NSHAPES; sh++)
        if (t->shape_ms[sh] < t->shape_ms[best]) best = sh;
    t->shape = best;
    return CSX_OK;
}

int gaxpy_tiled_run(const Csc *A, const double *x, double *y) { return gaxpy_tiled_launch(A, x, y, A->tiled->shape); }

static int gaxpy_tiled_launch(const Csc *A, const double *x, double *y, int shape) {
    const TiledPlan *t = A->tiled;
    hipStream_t s = ctx().stream;
    const size_t lds = (((size_t)(t->row_block + 1) * sizeof(double)) + 15) & ~(size_t)15;  // + dummy row
    const int32_t nwg = ctx().cus > 0 ? ctx().cus : 256;
    const unsigned grid = (unsigned)(t->nrb < nwg ? t->nrb : nwg);
#define CSX_TILED_LAUNCH_K(V, NW, NG, K24)                                                                           \
    {                                                                                                                \
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gaxpy_tiled<V, NW, NG, K24>),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES));                      \
        hipLaunchKernelGGL((k_gaxpy_tiled<V, NW, NG, K24>), dim3(grid), dim3(64 * NW), lds, s, t->tile_ptr,          \
                           (const uint32_t *)t->tile_len,                                                            \
                           K24 ? reinterpret_cast<const uint32_t *>(t->tile_key24) : t->tile_key, t->tile_base,      \
                           t->tile_val, x, y, A->m, t->nrb, t->row_block, t->slab_cols, t->rb_bits);                 \
    }
#define CSX_TILED_LAUNCH(V, NW, NG)                                                                                  \
    {                                                                                                                \
        if (t->tile_key24) CSX_TILED_LAUNCH_K(V, NW, NG, true) else CSX_TILED_LAUNCH_K(V, NW, NG, false)             \
    }
#ifdef CSX_ABLATION
    // CSX_TILED_VARIANT = V + 100 * NW + 10000 * NG (NW, NG default to the shipped TL_WAVES, TL_NG)
    int code = 0;
    if (const char *e = std::getenv("CSX_TILED_VARIANT")) code = std::atoi(e);
    const int v = code % 100, nw = (code / 100) % 100 ? (code / 100) % 100 : TL_WAVES, ng = code / 10000 ? code / 10000 : TL_NG;
#define CSX_V(V)                                             \
    if (v == V) {                                            \
        if (nw == 16 && ng == 2) CSX_TILED_LAUNCH(V, 16, 2)  \
        else if (nw == 4 && ng == 5) CSX_TILED_LAUNCH(V, 4, 5) \
        else if (nw == 2 && ng == 10) CSX_TILED_LAUNCH(V, 2, 10) \
        else return CSX_EINVAL;                              \
    } else
    if (v == 0) {
        if (nw == 16 && ng == 1) CSX_TILED_LAUNCH(0, 16, 1)
        else if (nw == 16 && ng == 2) CSX_TILED_LAUNCH(0, 16, 2)
        else if (nw == 12 && ng == 2) CSX_TILED_LAUNCH(0, 12, 2)
        else if (nw == 12 && ng == 4) CSX_TILED_LAUNCH(0, 12, 4)
        else if (nw == 8 && ng == 2) CSX_TILED_LAUNCH(0, 8, 2)
        else if (nw == 8 && ng == 3) CSX_TILED_LAUNCH(0, 8, 3)
        else if (nw == 8 && ng == 4) CSX_TILED_LAUNCH(0, 8, 4)
        else if (nw == 4 && ng == 4) CSX_TILED_LAUNCH(0, 4, 4)
        else if (nw == 4 && ng == 5) CSX_TILED_LAUNCH(0, 4, 5)
        else if (nw == 4 && ng == 6) CSX_TILED_LAUNCH(0, 4, 6)
        else if (nw == 2 && ng == 6) CSX_TILED_LAUNCH(0, 2, 6)
        else if (nw == 2 && ng == 7) CSX_TILED_LAUNCH(0, 2, 7)
        else if (nw == 2 && ng == 8) CSX_TILED_LAUNCH(0, 2, 8)
        else if (nw == 2 && ng == 9) CSX_TILED_LAUNCH(0, 2, 9)
        else if (nw == 2 && ng == 10) CSX_TILED_LAUNCH(0, 2, 10)
        else if (nw == 2 && ng == 12) CSX_TILED_LAUNCH(0, 2, 12)
        else if (nw == 1 && ng == 10) CSX_TILED_LAUNCH(0, 1, 10)
        else if (nw == 1 && ng == 12) CSX_TILED_LAUNCH(0, 1, 12)
        else if (nw == 3 && ng == 5) CSX_TILED_LAUNCH(0, 3, 5)
        else if (nw == 3 && ng == 6) CSX_TILED_LAUNCH(0, 3, 6)
        else if (nw == 3 && ng == 7) CSX_TILED_LAUNCH(0, 3, 7)
        else if (nw == 3 && ng == 8) CSX_TILED_LAUNCH(0, 3, 8)
        else if (nw == 5 && ng == 4) CSX_TILED_LAUNCH(0, 5, 4)
        else if (nw == 6 && ng == 3) CSX_TILED_LAUNCH(0, 6, 3)
        else if (nw == 4 && ng == 7) CSX_TILED_LAUNCH(0, 4, 7)
        else return CSX_EINVAL;
    } else
    CSX_V(1) CSX_V(2) CSX_V(3) CSX_V(4) CSX_V(5) CSX_V(6) CSX_V(7) return CSX_EINVAL;
#undef CSX_V
    (void)shape;
#else
    switch (shape) {
        case 1: CSX_TILED_LAUNCH(0, 2, 10) break;
        case 2: CSX_TILED_LAUNCH(0, 8, 4) break;
        case 3: CSX_TILED_LAUNCH(0, 2, 8) break;
        default: CSX_TILED_LAUNCH(0, TL_WAVES, TL_NG) break;     // 0 and -1 (not tuned): 4 x 5
    }
#endif
#undef CSX_TILED_LAUNCH
#undef CSX_TILED_LAUNCH_K
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx
