// cs_multiply + cs_scatter (csparse.py:1608-1642, :1961-1989): C = A * B.
//
// The reference builds column j of C by walking B(:,j) in storage order and,
// inside it, A(:,Bi[p]) in storage order (cs_scatter), appending a row to C the
// first time it is touched and summing later products into it.  So the PATTERN
// of a column is in first-touch order of that product sequence, and i[] must
// match it bit for bit.
//
// Device algorithm, one workgroup per column of C, two passes:
//   number the products of column j  t = 0, 1, 2, ...  in the reference's order;
//   pass A: every product (t, row, value) goes into an accumulator keyed by row:
//           tmin[row] = min(tmin[row], t)   (atomicMin)
//           val[row] += value               (ds_add_f64 / global atomic)
//   pass B: walk the products again; product t is the first touch of its row iff
//           tmin[row] == t; its position in the column is the number of first
//           touches before it (workgroup prefix count over t).  Emit (row, val[row]).
// The symbolic run is pass A alone, counting rows whose tmin was still unset; an
// exclusive scan of the counts gives C.p exactly, then the numeric run fills
// C.i / C.x in place.  Numerically cancelled entries are kept, like the reference.
//
// Accumulator kinds, chosen per column by its number of products P and by m:
//   LDS dense   m <= 8192: tmin/val arrays indexed by row in LDS (the reference's
//               w[] / x[] workspace, per workgroup)
//   LDS hash    open addressing, 1024..8192 slots, load <= 1/2
//   global dense  per-workgroup w[]/x[] in HBM scratch for columns too big for LDS
//
// Values are summed in arrival order (not the reference's), so x[] agrees to
// rounding (<= 1e-10 relative), p[] and i[] exactly.
#include <algorithm>

#include "csx_internal.h"

namespace csx {

constexpr int SG_THREADS = 256;
constexpr int SG_SEG = 1024;          // entries of B(:,j) staged per segment
constexpr uint32_t SG_UNSET = 0xFFFFFFFFu;
constexpr int SG_DENSE_MAX = 8192;    // rows for the LDS dense accumulator
constexpr int SG_GLOBAL_WGS = 64;     // concurrent workgroups with global accumulators

enum { ACC_LDS_DENSE = 0, ACC_LDS_HASH = 1, ACC_GLOBAL_DENSE = 2 };

__global__ __launch_bounds__(256) void k_sg_products(int32_t n, const int32_t *__restrict__ Ap,
                                                     const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                     int32_t m, uint32_t *__restrict__ bin, uint32_t *__restrict__ colid,
                                                     int32_t *__restrict__ hprod, unsigned long long *too_big) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    unsigned long long P = 0;
    for (int32_t p = Bp[j] + lane; p < Bp[j + 1]; p += 64) {
        const int32_t c = Bi[p];
        P += (unsigned long long)(Ap[c + 1] - Ap[c]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) P += __shfl_xor(P, d, 64);
    if (lane == 0) {
        if (P > 0xFFFFFFF0ull) atomicAdd(too_big, 1ull);
        const unsigned long long bound = P < (unsigned long long)m ? P : (unsigned long long)m;
        uint32_t b;
        if (P == 0) b = 7;                       // nothing to do
        else if (m <= SG_DENSE_MAX) b = 0;       // LDS dense
        else if (bound <= 512) b = 1;            // hash 1024
        else if (bound <= 1024) b = 2;           // hash 2048
        else if (bound <= 2048) b = 3;           // hash 4096
        else if (bound <= 4096) b = 4;           // hash 8192
        else b = 5;                              // global dense
        bin[j] = b;
        colid[j] = (uint32_t)j;
        const bool hashed = b >= 1 && b <= 4;
        hprod[j] = hashed ? (int32_t)P : 0;      // one-pass path: slots reserved in the product-order buffer
        if (hashed) atomicAdd(too_big + 1, P);
    }
}

struct Acc {
    int kind;
    uint32_t mask;       // hash: slots - 1
    int shift;           // hash: 32 - log2(slots)
    uint32_t *keys;      // hash only
    uint32_t *tmin;
    double *val;
};

__device__ __forceinline__ uint32_t acc_slot(const Acc &a, uint32_t row) {
    if (a.kind != ACC_LDS_HASH) return row;
    uint32_t s = (row * 0x9E3779B1u) >> a.shift;
    for (;;) {
        const uint32_t k = a.keys[s];
        if (k == row) return s;
        if (k == SG_UNSET) {
            const uint32_t prev = atomicCAS(&a.keys[s], SG_UNSET, row);
            if (prev == SG_UNSET || prev == row) return s;
        }
        s = (s + 1) & a.mask;
    }
}

__device__ __forceinline__ uint32_t acc_find(const Acc &a, uint32_t row) {
    if (a.kind != ACC_LDS_HASH) return row;
    uint32_t s = (row * 0x9E3779B1u) >> a.shift;
    while (a.keys[s] != row) s = (s + 1) & a.mask;
    return s;
}

// 256-thread exclusive prefix count of a flag; returns the prefix, *total = flags in the workgroup
__device__ __forceinline__ int prefix_count(bool flag, int *wsum, int *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int in_wave = __popcll(bal & lt);
    __syncthreads();
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SG_THREADS / 64; k++) {
        if (k < w) off += wsum[k];
        tot += wsum[k];
    }
    *total = tot;
    return off + in_wave;
}

// One workgroup processes columns cols[blockIdx.x], cols[blockIdx.x + gridDim.x], ...
template <bool NUMERIC, bool VALUES>
__global__ __launch_bounds__(SG_THREADS) void k_spgemm(int kind, int slots, int32_t m, const uint32_t *__restrict__ cols,
                                                       int32_t ncols, const int32_t *__restrict__ Ap,
                                                       const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                       const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                       const double *__restrict__ Bx, int32_t *__restrict__ count,
                                                       const int32_t *__restrict__ Cp, int32_t *__restrict__ Ci,
                                                       double *__restrict__ Cx, uint32_t *g_tmin, double *g_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [seg_off u32 x (SEG+1)] [seg_col i32 x SEG] [seg_bx f64 x SEG] [wsum, misc i32 x 16] [accumulator]
    uint32_t *seg_off = reinterpret_cast<uint32_t *>(smem);
    int32_t *seg_col = reinterpret_cast<int32_t *>(seg_off + SG_SEG + 4);
    double *seg_bx = reinterpret_cast<double *>(seg_col + SG_SEG);
    int *misc = reinterpret_cast<int *>(seg_bx + SG_SEG);
    unsigned char *accmem = reinterpret_cast<unsigned char *>(misc + 16);
    Acc a;
    a.kind = kind;
    a.mask = 0;
    a.shift = 0;
    a.keys = nullptr;
    int nacc;  // accumulator entries to clear per column
    if (kind == ACC_LDS_HASH) {
        a.mask = (uint32_t)slots - 1u;
        a.shift = 32 - (31 - __clz(slots));
        a.val = reinterpret_cast<double *>(accmem);
        a.tmin = reinterpret_cast<uint32_t *>(a.val + slots);
        a.keys = a.tmin + slots;
        nacc = slots;
    } else if (kind == ACC_LDS_DENSE) {
        a.val = reinterpret_cast<double *>(accmem);
        a.tmin = reinterpret_cast<uint32_t *>(a.val + ((m + 1) & ~1));
        nacc = m;
    } else {
        a.tmin = g_tmin + (size_t)blockIdx.x * (size_t)m;
        a.val = g_val ? g_val + (size_t)blockIdx.x * (size_t)m : nullptr;
        nacc = 0;  // global accumulators are kept clean by resetting touched rows
    }
    for (int32_t ci = blockIdx.x; ci < ncols; ci += gridDim.x) {
        const int32_t j = (int32_t)cols[ci];
        for (int k = threadIdx.x; k < nacc; k += SG_THREADS) {
            a.tmin[k] = SG_UNSET;
            if (NUMERIC && VALUES) a.val[k] = 0.0;
            if (kind == ACC_LDS_HASH) a.keys[k] = SG_UNSET;
        }
        if (threadIdx.x == 0) misc[8] = 0;  // distinct-row counter (symbolic)
        __syncthreads();
        const int32_t bb = Bp[j], be = Bp[j + 1];
        // ---- pass A (pass == 0) then, if NUMERIC, pass B (pass == 1), then reset for global ----
        const int npass = NUMERIC ? (kind == ACC_GLOBAL_DENSE ? 3 : 2) : (kind == ACC_GLOBAL_DENSE ? 2 : 1);
        for (int pass = 0; pass < npass; pass++) {
            const bool reset_pass = (kind == ACC_GLOBAL_DENSE) && pass == npass - 1;
            uint32_t tbase = 0;       // products before this segment
            int out_base = 0;         // first touches emitted so far (pass B)
            for (int32_t s0 = bb; s0 < be; s0 += SG_SEG) {
                const int nseg = min(SG_SEG, be - s0);
                __syncthreads();
                // stage the segment: column ids, B values, lengths -> exclusive offsets
                for (int k = threadIdx.x; k < nseg; k += SG_THREADS) {
                    const int32_t c = Bi[s0 + k];
                    const int32_t ab = Ap[c];
                    seg_col[k] = ab;  // start of A(:, c): saves a dependent load per product
                    seg_off[k] = (uint32_t)(Ap[c + 1] - ab);
                    if (VALUES) seg_bx[k] = Bx[s0 + k];
                }
                __syncthreads();
                if (threadIdx.x < 64) {  // one wave scans the (<= 1024) lengths, 16 per lane
                    const int lane = threadIdx.x;
                    uint32_t loc[SG_SEG / 64];
                    uint32_t sum = 0;
#pragma unroll
                    for (int k = 0; k < SG_SEG / 64; k++) {
                        const int idx = lane * (SG_SEG / 64) + k;
                        loc[k] = idx < nseg ? seg_off[idx] : 0u;
                        sum += loc[k];
                    }
                    uint32_t inc = sum;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        uint32_t t = __shfl_up(inc, d, 64);
                        if (lane >= d) inc += t;
                    }
                    uint32_t run = inc - sum;
#pragma unroll
                    for (int k = 0; k < SG_SEG / 64; k++) {
                        const int idx = lane * (SG_SEG / 64) + k;
                        if (idx < nseg) seg_off[idx] = run;
                        run += loc[k];
                    }
                    if (lane == 63) seg_off[SG_SEG] = inc;  // products in this segment
                }
                __syncthreads();
                const uint32_t nprod = seg_off[SG_SEG];
                // four 256-wide chunks of products per step: their (row, value) loads are issued
                // back to back (clamped indices, so no branch), then consumed in product order
                constexpr int UN = 4;
                for (uint32_t c0 = 0; c0 < nprod; c0 += UN * SG_THREADS) {
                    uint32_t rows_[UN];
                    double prods_[UN];
#pragma unroll
                    for (int u = 0; u < UN; u++) {
                        uint32_t tl = c0 + u * SG_THREADS + threadIdx.x;
                        if (tl >= nprod) tl = nprod - 1;
                        // which entry of the segment owns product tl: last k with seg_off[k] <= tl
                        int lo = 0, hi = nseg - 1;
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (seg_off[mid] <= tl) lo = mid;
                            else hi = mid - 1;
                        }
                        const int32_t q = seg_col[lo] + (int32_t)(tl - seg_off[lo]);
                        rows_[u] = (uint32_t)Ai[q];
                        prods_[u] = (NUMERIC && VALUES && pass == 0) ? seg_bx[lo] * Ax[q] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < UN; u++) {
                        const uint32_t tl = c0 + u * SG_THREADS + threadIdx.x;
                        if (c0 + u * SG_THREADS >= nprod) break;  // uniform
                        const bool live = tl < nprod;
                        const uint32_t row = rows_[u];
                        const double prod = prods_[u];
                        const uint32_t t = tbase + tl;
                        if (reset_pass) {
                            if (live) {
                                a.tmin[row] = SG_UNSET;
                                if (NUMERIC && VALUES) a.val[row] = 0.0;
                            }
                        } else if (pass == 0) {
                            if (live) {
                                const uint32_t slot = acc_slot(a, row);
                                const uint32_t old = atomicMin(&a.tmin[slot], t);
                                if (!NUMERIC && old == SG_UNSET) atomicAdd(&misc[8], 1);
                                if (NUMERIC && VALUES) unsafeAtomicAdd(&a.val[slot], prod);
                            }
                        } else {  // pass B
                            uint32_t slot = 0;
                            bool first = false;
                            if (live) {
                                slot = acc_find(a, row);
                                first = a.tmin[slot] == t;
                            }
                            int tot;
                            const int pos = out_base + prefix_count(first, misc, &tot);
                            if (first) {
                                Ci[Cp[j] + pos] = (int32_t)row;
                                if (VALUES) Cx[Cp[j] + pos] = a.val[slot];
                            }
                            out_base += tot;
                        }
                    }
                }
                tbase += nprod;
            }
            __syncthreads();
        }
        if (!NUMERIC && threadIdx.x == 0) count[j] = misc[8];
        __syncthreads();
    }
}

// ---- one-pass kernel for the LDS-hash bins ---------------------------------------------------------
// Columns whose products fit an LDS hash table (P <= 4096, m > 8192) are finished in ONE walk over
// their products: every product (t, row, value) is inserted (CAS on the key, ds_min on tmin, ds_add_f64
// on the value); afterwards the first-touch ORDER is recovered without walking the products again:
// a bitmap over t marks the tmin of every occupied slot, and the position of a row in the reference's
// column is the number of marked bits below its tmin (prefix popcount).  Rows and sums go to a
// product-order buffer at the column's upper-bound offset (exclusive scan of P); once the exact counts
// are scanned into C.p a streaming kernel compacts them into C.i / C.x.
constexpr int H1_SEG = 256;     // entries of B(:,j) staged per segment
constexpr int H1_MAXP = 4096;   // products per column (bitmap bits)
constexpr int H1_UN = 4;        // B entries in flight per 32-lane group

template <bool VALUES>
__device__ __forceinline__ void h1_insert(uint32_t *keys, uint32_t *tmin, double *val, uint32_t mask, int shift,
                                          uint32_t row, uint32_t t, double v) {
    uint32_t s = (row * 0x9E3779B1u) >> shift;
    for (;;) {
        const uint32_t prev = atomicCAS(&keys[s], SG_UNSET, row);
        if (prev == SG_UNSET || prev == row) break;
        s = (s + 1) & mask;
    }
    atomicMin(&tmin[s], t);
    if (VALUES) unsafeAtomicAdd(&val[s], v);
}

template <bool VALUES>
__global__ __launch_bounds__(256) void k_sg_hash1(int slots, const uint32_t *__restrict__ cols, int32_t ncols,
                                                  const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                  const double *__restrict__ Ax, const int32_t *__restrict__ Bp,
                                                  const int32_t *__restrict__ Bi, const double *__restrict__ Bx,
                                                  const int32_t *__restrict__ toff, int32_t *__restrict__ count,
                                                  int32_t *__restrict__ tmp_i, double *__restrict__ tmp_x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *val = reinterpret_cast<double *>(smem);
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + (VALUES ? (size_t)slots * 8 : 0));
    uint32_t *tmin = keys + slots;
    double *seg_bx = reinterpret_cast<double *>(tmin + slots);
    int32_t *seg_ab = reinterpret_cast<int32_t *>(seg_bx + H1_SEG);
    uint32_t *seg_len = reinterpret_cast<uint32_t *>(seg_ab + H1_SEG);
    uint32_t *seg_off = seg_len + H1_SEG;
    uint32_t *bitmap = seg_off + H1_SEG;        // H1_MAXP / 32 words
    uint32_t *wpre = bitmap + H1_MAXP / 32;     // exclusive popcount prefix per word
    uint32_t *misc = wpre + H1_MAXP / 32;       // [0..3] wave sums, [4] segment products, [5] column count
    const uint32_t mask = (uint32_t)slots - 1u;
    const int shift = 32 - (31 - __clz(slots));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int grp = tid >> 5, gl = tid & 31;
    for (int k = tid; k < slots; k += 256) {
        keys[k] = SG_UNSET;
        tmin[k] = SG_UNSET;
        if (VALUES) val[k] = 0.0;
    }
    if (tid < H1_MAXP / 32) bitmap[tid] = 0u;
    __syncthreads();
    for (int32_t ci = blockIdx.x; ci < ncols; ci += gridDim.x) {
        const int32_t j = (int32_t)cols[ci];
        const int32_t bb = Bp[j], be = Bp[j + 1];
        uint32_t tbase = 0;
        for (int32_t s0 = bb; s0 < be; s0 += H1_SEG) {
            const int nseg = min(H1_SEG, be - s0);
            // stage the segment and scan the lengths of the A columns it names
            uint32_t len = 0;
            if (tid < nseg) {
                const int32_t c = Bi[s0 + tid];
                const int32_t ab = Ap[c];
                len = (uint32_t)(Ap[c + 1] - ab);
                seg_ab[tid] = ab;
                seg_len[tid] = len;
                if (VALUES) seg_bx[tid] = Bx[s0 + tid];
            }
            uint32_t inc = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            if (lane == 63) misc[wv] = inc;
            __syncthreads();
            uint32_t woff = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < wv) woff += misc[k];
            seg_off[tid] = woff + inc - len;
            if (tid == 255) misc[4] = woff + inc;
            __syncthreads();
            // 32 lanes per entry of B(:,j); H1_UN entries in flight per group
            for (int k0 = grp; k0 < nseg; k0 += 8 * H1_UN) {
                uint32_t rows_[H1_UN], ts_[H1_UN], lens_[H1_UN];
                int32_t abs_[H1_UN];
                double vs_[H1_UN], bxs_[H1_UN];
#pragma unroll
                for (int u = 0; u < H1_UN; u++) {
                    const int k = k0 + 8 * u;
                    const bool have = k < nseg;
                    const int kk = have ? k : k0;
                    abs_[u] = seg_ab[kk];
                    lens_[u] = have ? seg_len[kk] : 0u;
                    ts_[u] = tbase + seg_off[kk] + (uint32_t)gl;
                    bxs_[u] = VALUES ? seg_bx[kk] : 0.0;
                    const int32_t q = (uint32_t)gl < lens_[u] ? abs_[u] + gl : 0;   // clamped: nnz(A) > 0
                    rows_[u] = (uint32_t)Ai[q];
                    vs_[u] = VALUES ? Ax[q] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < H1_UN; u++) {
                    if ((uint32_t)gl < lens_[u]) h1_insert<VALUES>(keys, tmin, val, mask, shift, rows_[u], ts_[u], bxs_[u] * vs_[u]);
                    for (uint32_t q = (uint32_t)gl + 32; q < lens_[u]; q += 32) {   // A columns longer than 32
                        const uint32_t row = (uint32_t)Ai[abs_[u] + (int32_t)q];
                        const double v = VALUES ? bxs_[u] * Ax[abs_[u] + (int32_t)q] : 0.0;
                        h1_insert<VALUES>(keys, tmin, val, mask, shift, row, ts_[u] + (q - (uint32_t)gl), v);
                    }
                }
            }
            tbase += misc[4];
            __syncthreads();
        }
        // ---- read-out: mark first touches in product order, rank them, emit, and leave the table clean ----
        for (int s = tid; s < slots; s += 256)
            if (keys[s] != SG_UNSET) {
                const uint32_t t = tmin[s];
                atomicOr(&bitmap[t >> 5], 1u << (t & 31));
            }
        __syncthreads();
        if (tid < 64) {
            const uint32_t c0 = __popc(bitmap[2 * tid]), c1 = __popc(bitmap[2 * tid + 1]);
            uint32_t inc = c0 + c1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (tid >= d) inc += up;
            }
            wpre[2 * tid] = inc - c0 - c1;
            wpre[2 * tid + 1] = inc - c1;
            if (tid == 63) misc[5] = inc;
        }
        __syncthreads();
        const int64_t base = toff[j];
        for (int s = tid; s < slots; s += 256) {
            const uint32_t key = keys[s];
            if (key != SG_UNSET) {
                const uint32_t t = tmin[s];
                const uint32_t pos = wpre[t >> 5] + __popc(bitmap[t >> 5] & ((1u << (t & 31)) - 1u));
                tmp_i[base + pos] = (int32_t)key;
                keys[s] = SG_UNSET;
                tmin[s] = SG_UNSET;
                if (VALUES) {
                    tmp_x[base + pos] = val[s];
                    val[s] = 0.0;
                }
            }
        }
        if (tid == 0) count[j] = (int32_t)misc[5];
        __syncthreads();
        if (tid < H1_MAXP / 32) bitmap[tid] = 0u;
        // the next column's first use of bitmap / misc / seg_* is behind the staging barriers above
    }
}

// one wave per column: move its rows (and sums) from the product-order buffer to their place in C
__global__ __launch_bounds__(256) void k_sg_compact(const uint32_t *__restrict__ cols, int32_t ncols,
                                                    const int32_t *__restrict__ toff, const int32_t *__restrict__ Cp,
                                                    const int32_t *__restrict__ tmp_i, const double *__restrict__ tmp_x,
                                                    int32_t *__restrict__ Ci, double *__restrict__ Cx) {
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= ncols) return;
    const int32_t j = (int32_t)cols[w];
    const int64_t src = toff[j], dst = Cp[j];
    const int32_t cnt = Cp[j + 1] - Cp[j];
    for (int32_t k = lane; k < cnt; k += 64) {
        Ci[dst + k] = tmp_i[src + k];
        if (tmp_x) Cx[dst + k] = tmp_x[src + k];
    }
}

static size_t h1_lds_bytes(int slots, bool values) {
    return (size_t)slots * (values ? 16 : 8) + H1_SEG * (8 + 4 + 4 + 4) + (H1_MAXP / 32) * 8 + 64;
}

template <bool VALUES>
static int launch_hash1(int slots, const Csc *A, const Csc *B, const uint32_t *cols, int32_t ncols, const int32_t *toff,
                        int32_t *count, int32_t *tmp_i, double *tmp_x) {
    if (ncols <= 0) return CSX_OK;
    const size_t lds = h1_lds_bytes(slots, VALUES);
    auto kern = k_sg_hash1<VALUES>;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / (int64_t)(lds + 512)));
    const int64_t grid = std::min<int64_t>(ncols, (int64_t)ctx().cus * per_cu * 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx().stream, slots, cols, ncols, A->p, A->i, A->x, B->p,
                       B->i, B->x, toff, count, tmp_i, tmp_x);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *p, int64_t n, uint32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

static size_t sg_lds_bytes(int kind, int slots, int32_t m) {
    size_t base = (SG_SEG + 4) * 4 + SG_SEG * 4 + SG_SEG * 8 + 16 * 4;
    if (kind == ACC_LDS_HASH) return base + (size_t)slots * 16;
    if (kind == ACC_LDS_DENSE) return base + (size_t)((m + 1) & ~1) * 8 + (size_t)m * 4 + 16;
    return base;
}

template <bool NUMERIC, bool VALUES>
static int launch_bin(int kind, int slots, const Csc *A, const Csc *B, const uint32_t *cols, int32_t ncols,
                      int32_t *count, const int32_t *Cp, int32_t *Ci, double *Cx, uint32_t *g_tmin, double *g_val) {
    if (ncols <= 0) return CSX_OK;
    const size_t lds = sg_lds_bytes(kind, slots, A->m);
    auto kern = k_spgemm<NUMERIC, VALUES>;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    int64_t grid = ncols;
    if (kind == ACC_GLOBAL_DENSE) grid = std::min<int64_t>(grid, SG_GLOBAL_WGS);
    else grid = std::min<int64_t>(grid, (int64_t)ctx().cus * 16);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SG_THREADS), lds, ctx().stream, kind, slots, A->m, cols, ncols,
                       A->p, A->i, A->x, B->p, B->i, B->x, count, Cp, Ci, Cx, g_tmin, g_val);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

template <bool NUMERIC, bool VALUES>
static int run_bins(const Csc *A, const Csc *B, const uint32_t *cols, const int32_t *bin_ptr, int32_t *count,
                    const int32_t *Cp, int32_t *Ci, double *Cx, uint32_t *g_tmin, double *g_val, bool skip_hash) {
    static const int slots_of_bin[5] = {0, 1024, 2048, 4096, 8192};
    for (int b = 0; b <= 5; b++) {
        if (skip_hash && b >= 1 && b <= 4) continue;  // done by the one-pass kernel
        const int32_t nb = bin_ptr[b + 1] - bin_ptr[b];
        const int kind = b == 0 ? ACC_LDS_DENSE : (b == 5 ? ACC_GLOBAL_DENSE : ACC_LDS_HASH);
        CSX_TRY((launch_bin<NUMERIC, VALUES>(kind, b >= 1 && b <= 4 ? slots_of_bin[b] : 0, A, B, cols + bin_ptr[b], nb,
                                             count, Cp, Ci, Cx, g_tmin, g_val)));
    }
    return CSX_OK;
}

static int multiply_device(const Csc *A, const Csc *B, Csc *C) {
    hipStream_t s = ctx().stream;
    const int32_t m = A->m, n = B->n;
    const bool values = A->x && B->x;
    C->m = m;
    C->n = n;
    C->owns = true;
    CSX_TRY(dalloc(&C->p, (size_t)n + 1));
    if (n == 0 || m == 0 || A->nnz == 0 || B->nnz == 0) {
        CSX_HIP(hipMemsetAsync(C->p, 0, ((size_t)n + 1) * sizeof(int32_t), s));
        C->nnz = 0;
        CSX_TRY(dalloc(&C->i, 0));
        if (values) CSX_TRY(dalloc(&C->x, 0));
        return CSX_OK;
    }
    uint32_t *bin = nullptr, *colid = nullptr, *sbin = nullptr, *scol = nullptr, *g_tmin = nullptr;
    int32_t *bin_ptr_d = nullptr, *count = nullptr, *hprod = nullptr, *toff = nullptr, *tmp_i = nullptr;
    double *g_val = nullptr, *tmp_x = nullptr;
    unsigned long long *too_big = nullptr;  // [0] columns with >= 2^32 products, [1] products in hash-bin columns
    int st = dalloc(&bin, (size_t)n);
    if (st == CSX_OK) st = dalloc(&colid, (size_t)n);
    if (st == CSX_OK) st = dalloc(&sbin, (size_t)n);
    if (st == CSX_OK) st = dalloc(&scol, (size_t)n);
    if (st == CSX_OK) st = dalloc(&bin_ptr_d, 9);
    if (st == CSX_OK) st = dalloc(&count, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&hprod, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&too_big, 2);
    int32_t bin_ptr[9] = {0};
    unsigned long long big[2] = {0, 0};
    if (st == CSX_OK) {
        (void)hipMemsetAsync(too_big, 0, 2 * sizeof(unsigned long long), s);
        (void)hipMemsetAsync(count, 0, ((size_t)n + 1) * sizeof(int32_t), s);
        hipLaunchKernelGGL(k_sg_products, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, A->p, B->p, B->i, m,
                           bin, colid, hprod, too_big);
        st = stable_sort_by_key(bin, colid, nullptr, n, 8, sbin, scol, nullptr);
    }
    if (st == CSX_OK) st = boundaries_from_sorted(sbin, n, 8, bin_ptr_d);
    if (st == CSX_OK) {
        if (hipMemcpyAsync(bin_ptr, bin_ptr_d, 9 * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(big, too_big, sizeof big, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    if (st == CSX_OK && big[0]) st = CSX_EINVAL;  // a column with >= 2^32 products
    // one-pass path for the hash bins when its product-order buffer (12 B per product) is affordable
    const int32_t nhash = bin_ptr[5] - bin_ptr[1];
    bool onepass = false;
    if (st == CSX_OK && nhash > 0 && big[1] < 0x7FFFFFF0ull && !getenv("CSX_SPGEMM_TWO_PASS")) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)big[1] * (values ? 12 : 4);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 3) onepass = true;
    }
    if (st == CSX_OK && onepass) {
        st = dalloc(&toff, (size_t)n + 1);
        int64_t tot = 0;
        if (st == CSX_OK) st = scan_exclusive_i32(hprod, toff, n, &tot);
        if (st == CSX_OK) st = dalloc(&tmp_i, (size_t)big[1]);
        if (st == CSX_OK && values) st = dalloc(&tmp_x, (size_t)big[1]);
        static const int slots_of_bin[5] = {0, 1024, 2048, 4096, 8192};
        for (int b = 1; b <= 4 && st == CSX_OK; b++) {
            const int32_t nb = bin_ptr[b + 1] - bin_ptr[b];
            st = values ? launch_hash1<true>(slots_of_bin[b], A, B, scol + bin_ptr[b], nb, toff, count, tmp_i, tmp_x)
                        : launch_hash1<false>(slots_of_bin[b], A, B, scol + bin_ptr[b], nb, toff, count, tmp_i, nullptr);
        }
    }
    const int32_t nglobal = bin_ptr[6] - bin_ptr[5];
    if (st == CSX_OK && nglobal > 0) {
        const size_t wgs = (size_t)std::min<int32_t>(nglobal, SG_GLOBAL_WGS);
        st = dalloc(&g_tmin, wgs * (size_t)m);
        if (st == CSX_OK && values) st = dalloc(&g_val, wgs * (size_t)m);
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, s, g_tmin, (int64_t)(wgs * (size_t)m), SG_UNSET);
            if (values) (void)hipMemsetAsync(g_val, 0, wgs * (size_t)m * sizeof(double), s);
        }
    }
    // symbolic: distinct rows per column -> C.p
    if (st == CSX_OK)
        st = run_bins<false, false>(A, B, scol, bin_ptr, count, nullptr, nullptr, nullptr, g_tmin, nullptr, onepass);
    int64_t total = 0;
    if (st == CSX_OK) st = scan_exclusive_i32(count, C->p, n, &total);
    if (st == CSX_OK && total > 0x7FFFFFFFll) {
        set_error("cs_multiply: the product has %lld entries (int32 indices)", (long long)total);
        st = CSX_EINVAL;
    }
    if (st == CSX_OK) {
        C->nnz = (int32_t)total;
        st = dalloc(&C->i, (size_t)total);
        if (st == CSX_OK && values) st = dalloc(&C->x, (size_t)total);
    }
    if (st == CSX_OK && onepass) {
        hipLaunchKernelGGL(k_sg_compact, dim3((unsigned)(((int64_t)nhash + 3) / 4)), dim3(256), 0, s, scol + bin_ptr[1],
                           nhash, toff, C->p, tmp_i, tmp_x, C->i, C->x);
        if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
    }
    if (st == CSX_OK) {
        if (values) st = run_bins<true, true>(A, B, scol, bin_ptr, count, C->p, C->i, C->x, g_tmin, g_val, onepass);
        else st = run_bins<true, false>(A, B, scol, bin_ptr, count, C->p, C->i, nullptr, g_tmin, nullptr, onepass);
    }
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) {
        set_error("cs_multiply: %s", hipGetErrorString(hipGetLastError()));
        st = CSX_ERUNTIME;
    }
    dfree(hprod);
    dfree(toff);
    dfree(tmp_i);
    dfree(tmp_x);
    dfree(bin);
    dfree(colid);
    dfree(sbin);
    dfree(scol);
    dfree(bin_ptr_d);
    dfree(count);
    dfree(too_big);
    dfree(g_tmin);
    dfree(g_val);
    return st;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_multiply(csx_handle_t hA, csx_handle_t hB, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA), *B = csc(hB);
    if (!A || !B || !out || A->n != B->m) return CSX_EINVAL;
    Csc *C = new Csc();
    int st = multiply_device(A, B, C);
    if (st != CSX_OK) {
        free_csc(C);
        return st;
    }
    *out = put(K_CSC, C);
    return CSX_OK;
}
