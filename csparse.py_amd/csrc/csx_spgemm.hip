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

// column bins (sort key): dense LDS accumulator, eight LDS-hash bins by number of products, global accumulator
// and, for the one-pass kernels, six more hash bins for "narrow" columns: at most 64 entries in B(:,j), each naming a
// column of A with at most 32 entries (k_sg_hash2)
constexpr int SG_BIN_DENSE = 0, SG_BIN_HASH0 = 1, SG_HASH_BINS = 8, SG_BIN_NARROW0 = 9, SG_NARROW_BINS = 6,
              SG_BIN_GLOBAL = 15, SG_BIN_EMPTY = 31, SG_NBINS = 32;
constexpr int H2_MAXSEG = 64, H2_MAXLEN = 32, H2_MAXP = 2048;
constexpr int SG_HASH_MAXP = 4096;
__host__ __device__ constexpr int sg_hash_limit(int hb) {   // products a column of hash bin hb may have
    return hb == 0 ? 256 : hb == 1 ? 512 : hb == 2 ? 768 : hb == 3 ? 1024 : hb == 4 ? 1536 : hb == 5 ? 2048 : hb == 6 ? 3072 : 4096;
}

__global__ __launch_bounds__(256) void k_sg_products(int32_t n, const int32_t *__restrict__ Ap,
                                                     const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                     int32_t m, uint32_t *__restrict__ bin, uint32_t *__restrict__ colid,
                                                     int32_t *__restrict__ hprod, unsigned long long *too_big) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    unsigned long long P = 0;
    int32_t maxlen = 0;
    for (int32_t p = Bp[j] + lane; p < Bp[j + 1]; p += 64) {
        const int32_t c = Bi[p];
        const int32_t len = Ap[c + 1] - Ap[c];
        P += (unsigned long long)len;
        maxlen = max(maxlen, len);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        P += __shfl_xor(P, d, 64);
        maxlen = max(maxlen, __shfl_xor(maxlen, d, 64));
    }
    if (lane == 0) {
        if (P > 0xFFFFFFF0ull) atomicAdd(too_big, 1ull);
        uint32_t b;
        if (P == 0) b = SG_BIN_EMPTY;
        else if (m <= SG_DENSE_MAX) b = SG_BIN_DENSE;
        else if (P > (unsigned long long)SG_HASH_MAXP) b = SG_BIN_GLOBAL;
        else {
            int hb = 0;
            while ((unsigned long long)sg_hash_limit(hb) < P) hb++;
            const bool narrow = Bp[j + 1] - Bp[j] <= H2_MAXSEG && maxlen <= H2_MAXLEN && P <= (unsigned long long)H2_MAXP;
            b = (uint32_t)((narrow ? SG_BIN_NARROW0 : SG_BIN_HASH0) + hb);
        }
        bin[j] = b;
        colid[j] = (uint32_t)j;
        hprod[j] = (b >= SG_BIN_HASH0 && b < SG_BIN_NARROW0 + SG_NARROW_BINS) ? (int32_t)P : 0;
    }
}

// sum of a non-negative int32 array in 64 bits (few atomics: one per wave of a small grid)
__global__ __launch_bounds__(256) void k_sum_i32(const int32_t *__restrict__ v, int64_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
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
__device__ __forceinline__ void h1_insert(uint32_t *keys, uint32_t *tmin, double *val, uint32_t slots, uint32_t row,
                                          uint32_t t, double v) {
    uint32_t s = __umulhi(row * 0x9E3779B1u, slots);
    for (;;) {
        const uint32_t prev = atomicCAS(&keys[s], SG_UNSET, row);
        if (prev == SG_UNSET || prev == row) break;
        s = s + 1 == slots ? 0u : s + 1;
    }
    atomicMin(&tmin[s], t);
    if (VALUES) unsafeAtomicAdd(&val[s], v);
}

// info[k] = (column j, Bp[j], Bp[j+1], offset of the column in the product-order buffer), in bin order
__global__ __launch_bounds__(256) void k_sg_colinfo(const uint32_t *__restrict__ cols, int32_t ncols,
                                                    const int32_t *__restrict__ Bp, const int32_t *__restrict__ toff,
                                                    int4 *__restrict__ info) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int32_t j = (int32_t)cols[k];
    info[k] = make_int4(j, Bp[j], Bp[j + 1], toff[j]);
}

// Persistent workgroups, columns ci = blockIdx.x, + gridDim.x, ...  The loads a column needs before its
// products can be gathered (its descriptor, its B entries, the A column extents they name) are issued one and
// two columns ahead and ride along with the current column's gathers, so the only exposed round trip per
// column is the gather of A rows / values itself.
template <bool VALUES>
__global__ __launch_bounds__(256) void k_sg_hash1(int slots_, const int4 *__restrict__ info, int32_t ncols,
                                                  const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                  const double *__restrict__ Ax, const int32_t *__restrict__ Bi,
                                                  const double *__restrict__ Bx, int32_t *__restrict__ count,
                                                  int32_t *__restrict__ tmp_i, double *__restrict__ tmp_x, int abl) {
    // abl (timing experiments, ablation build only; results are wrong unless 0): 1 = no inserts, 2 = no read-out /
    // copy-out, 4 = rows made up instead of gathered
#ifndef CSX_ABLATION
    abl = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t slots = (uint32_t)slots_;
    double *val = reinterpret_cast<double *>(smem);
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + (VALUES ? (size_t)slots * 8 : 0));
    uint32_t *tmin = keys + slots;
    double *seg_bx = reinterpret_cast<double *>(tmin + slots + (slots & 1));
    int32_t *seg_ab = reinterpret_cast<int32_t *>(seg_bx + H1_SEG);
    uint32_t *seg_len = reinterpret_cast<uint32_t *>(seg_ab + H1_SEG);
    uint32_t *seg_off = seg_len + H1_SEG;
    uint32_t *bitmap = seg_off + H1_SEG;        // H1_MAXP / 32 words
    uint32_t *wpre = bitmap + H1_MAXP / 32;     // exclusive popcount prefix per word
    uint32_t *misc = wpre + H1_MAXP / 32;       // [0..3] wave sums, [4] segment products, [5] column count
    uint16_t *inv = reinterpret_cast<uint16_t *>(misc + 16);   // position in the column -> slot (<= limit entries)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int grp = tid >> 5, gl = tid & 31;
    const int32_t G = (int32_t)gridDim.x;
    const int4 none = make_int4(-1, 0, 0, 0);
    int32_t ci = blockIdx.x;                    // the host launches at most ncols workgroups
    int4 cur = info[ci];
    int4 nxt = ci + G < ncols ? info[ci + G] : none;
    int32_t ab0 = 0;
    uint32_t len0 = 0;
    double bx0 = 0.0;
    if (tid < min(H1_SEG, cur.z - cur.y)) {
        const int32_t c = Bi[cur.y + tid];
        ab0 = Ap[c];
        len0 = (uint32_t)(Ap[c + 1] - ab0);
        if (VALUES) bx0 = Bx[cur.y + tid];
    }
    for (uint32_t k = tid; k < slots; k += 256) {
        keys[k] = SG_UNSET;
        tmin[k] = SG_UNSET;
        if (VALUES) val[k] = 0.0;
    }
    if (tid < H1_MAXP / 32) bitmap[tid] = 0u;
    __syncthreads();
    for (;;) {
        // ---- prefetch: descriptor two columns ahead, B entries one column ahead ----
        const int4 nn = ci + 2 * G < ncols ? info[ci + 2 * G] : none;
        const int nseg1 = nxt.x >= 0 ? min(H1_SEG, nxt.z - nxt.y) : 0;
        int32_t c1 = 0;
        double bx1 = 0.0;
        if (tid < nseg1) {
            c1 = Bi[nxt.y + tid];
            if (VALUES) bx1 = Bx[nxt.y + tid];
        }
        const int32_t j = cur.x, bb = cur.y, be = cur.z;
        uint32_t tbase = 0;
        for (int32_t s0 = bb; s0 < be; s0 += H1_SEG) {
            const int nseg = min(H1_SEG, be - s0);
            // stage the segment (the first one was prefetched) and scan the lengths of the A columns it names
            int32_t ab = ab0;
            uint32_t len = len0;
            double bx = bx0;
            if (s0 != bb) {
                len = 0;
                if (tid < nseg) {
                    const int32_t c = Bi[s0 + tid];
                    ab = Ap[c];
                    len = (uint32_t)(Ap[c + 1] - ab);
                    if (VALUES) bx = Bx[s0 + tid];
                }
            }
            if (tid < nseg) {
                seg_ab[tid] = ab;
                seg_len[tid] = len;
                if (VALUES) seg_bx[tid] = bx;
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
                    if (abl & 4) {
                        rows_[u] = (uint32_t)q * 2654435761u >> 12;
                        vs_[u] = 1.0;
                    } else {
                        rows_[u] = (uint32_t)Ai[q];
                        vs_[u] = VALUES ? Ax[q] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < H1_UN; u++) {
                    if ((abl & 1) && rows_[u] != 0x12345u) continue;
                    if ((uint32_t)gl < lens_[u]) h1_insert<VALUES>(keys, tmin, val, slots, rows_[u], ts_[u], bxs_[u] * vs_[u]);
                    for (uint32_t q = (uint32_t)gl + 32; q < lens_[u]; q += 32) {   // A columns longer than 32
                        const uint32_t row = (uint32_t)Ai[abs_[u] + (int32_t)q];
                        const double v = VALUES ? bxs_[u] * Ax[abs_[u] + (int32_t)q] : 0.0;
                        h1_insert<VALUES>(keys, tmin, val, slots, row, ts_[u] + (q - (uint32_t)gl), v);
                    }
                }
            }
            tbase += misc[4];
            __syncthreads();
        }
        // ---- prefetch, second half: extents of the A columns the next column names ----
        int32_t ab1 = 0, e1 = 0;
        if (tid < nseg1) {
            ab1 = Ap[c1];
            e1 = Ap[c1 + 1];
        }
        // ---- read-out: mark first touches in product order, rank them, emit, and leave the table clean ----
        if (abl & 2) {
            for (uint32_t k = tid; k < slots; k += 256) {
                keys[k] = SG_UNSET;
                tmin[k] = SG_UNSET;
                if (VALUES) val[k] = 0.0;
            }
            if (tid == 0) count[j] = 0;
            __syncthreads();
            ci += G;
            if (ci >= ncols) break;
            cur = nxt;
            nxt = nn;
            ab0 = ab1;
            len0 = (uint32_t)(e1 - ab1);
            bx0 = bx1;
            continue;
        }
        for (uint32_t s = tid; s < slots; s += 256)
            if (keys[s] != SG_UNSET) {
                const uint32_t t = tmin[s];
                atomicOr(&bitmap[t >> 5], 1u << (t & 31));
            }
        __syncthreads();
        if (tid < 64) {
            const uint32_t c0 = __popc(bitmap[2 * tid]), c1w = __popc(bitmap[2 * tid + 1]);
            uint32_t inc = c0 + c1w;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(inc, d, 64);
                if (tid >= d) inc += up;
            }
            wpre[2 * tid] = inc - c0 - c1w;
            wpre[2 * tid + 1] = inc - c1w;
            if (tid == 63) misc[5] = inc;
        }
        __syncthreads();
        // inv[position] = slot, then a position-ordered (coalesced) copy-out that also clears the table
        for (uint32_t s = tid; s < slots; s += 256)
            if (keys[s] != SG_UNSET) {
                const uint32_t t = tmin[s];
                const uint32_t pos = wpre[t >> 5] + __popc(bitmap[t >> 5] & ((1u << (t & 31)) - 1u));
                inv[pos] = (uint16_t)s;
            }
        __syncthreads();
        const int64_t base = cur.w;
        const uint32_t cnt = misc[5];
        for (uint32_t pos = tid; pos < cnt; pos += 256) {
            const uint32_t s = inv[pos];
            tmp_i[base + pos] = (int32_t)keys[s];
            keys[s] = SG_UNSET;
            tmin[s] = SG_UNSET;
            if (VALUES) {
                tmp_x[base + pos] = val[s];
                val[s] = 0.0;
            }
        }
        if (tid == 0) count[j] = (int32_t)cnt;
        __syncthreads();
        if (tid < H1_MAXP / 32) bitmap[tid] = 0u;
        // the next column's first use of bitmap / misc / seg_* is behind the staging barriers above
        ci += G;
        if (ci >= ncols) break;
        cur = nxt;
        nxt = nn;
        ab0 = ab1;
        len0 = (uint32_t)(e1 - ab1);
        bx0 = bx1;
    }
}

// ---- one-pass kernel for narrow columns (at most 64 entries in B(:,j), A columns of at most 32 entries) -------------
// The same algorithm with the per-column machinery cut down to three barriers and its instruction count following
// the column's length (the general kernel above is bound by instruction issue, not by LDS or memory):
//  * every wave keeps the whole B column in registers (entry l in lane l, prefetched one column ahead as above), so the
//    offsets of the A columns in product order are one wave scan; a wave instruction covers two entries, k0 in its low
//    half and k0 + 1 in its high half with k0 wave-uniform, so what a half-wave needs of its entry comes by v_readlane:
//    nothing is staged in LDS;
//  * a thread issues the compare-and-swaps of its products together and resolves collisions afterwards (one returning
//    LDS atomic per product is the latency that matters);
//  * a thread remembers slot and product number of its (at most 8) products, so first touches are found by asking
//    tmin[slot] == t -- no walk over the table, no bitmap: a half-wave holds one entry's products in order, a ballot
//    gives the first touches before each lane and their count per entry, one wave scan over the 64 entry counts the
//    rest -- and each first touch writes row and sum straight to the product-order buffer (runs) and wipes its slot;
//  * slot i of a thread = entries 8 i .. 8 i + 7; slots at or past the column's length are skipped in every phase.
template <bool VALUES, int N>
__device__ __forceinline__ void h2_insert(uint32_t *keys, uint32_t *tmin, double *val, uint32_t slots,
                                          const uint32_t *row, const uint32_t *t, const double *v, const bool *act,
                                          uint32_t *s) {
    uint32_t prev[N];
#pragma unroll
    for (int u = 0; u < N; u++) s[u] = __umulhi(row[u] * 0x9E3779B1u, slots);
#pragma unroll
    for (int u = 0; u < N; u++) prev[u] = act[u] ? atomicCAS(&keys[s[u]], SG_UNSET, row[u]) : SG_UNSET;
#pragma unroll
    for (int u = 0; u < N; u++)
        if (act[u] && prev[u] != SG_UNSET && prev[u] != row[u]) {
            for (;;) {
                s[u] = s[u] + 1 == slots ? 0u : s[u] + 1;
                const uint32_t p = atomicCAS(&keys[s[u]], SG_UNSET, row[u]);
                if (p == SG_UNSET || p == row[u]) break;
            }
        }
#pragma unroll
    for (int u = 0; u < N; u++)
        if (act[u]) {
            atomicMin(&tmin[s[u]], t[u]);
            if (VALUES) unsafeAtomicAdd(&val[s[u]], v[u]);
        }
}

__device__ __forceinline__ int h2_pick(int v, int k0, bool hi) {   // v of lane k0 (low half-wave) / k0 + 1 (high)
    const int a = __builtin_amdgcn_readlane(v, k0), b = __builtin_amdgcn_readlane(v, k0 + 1);
    return hi ? b : a;
}

// the (at most 8) products of a thread in one column: slot i = entries 8 i .. 8 i + 7 of B(:,j), two per wave
struct H2Prod {
    uint32_t row[8], t[8];
    double v[8];
    bool act[8];
};

// Gather the products of a column whose entries sit one per lane (ab = start of the A column the entry names, len its
// length, off = its first product number, bx = the entry's value).  A wave instruction covers two entries: k0 in
// its low half and k0 + 1 in its high half, k0 wave-uniform, so the half-waves get their entry by v_readlane.
template <bool VALUES>
__device__ __forceinline__ void h2_fetch(H2Prod &P, int32_t ab, uint32_t len, uint32_t off, double bx, int nseg, int wu,
                                         bool hi, int gl, const int32_t *__restrict__ Ai,
                                         const double *__restrict__ Ax, int abl) {
    const int bxlo = __double2loint(bx), bxhi = __double2hiint(bx);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        P.act[i] = false;
        P.row[i] = P.t[i] = 0;
        P.v[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < 4 || (i < 6 && nseg > 32) || nseg > 48) {   // uniform: slots past the column's length are skipped
            const int k0 = 8 * i + 2 * wu;
            const int32_t abk = h2_pick(ab, k0, hi);
            const uint32_t lenk = (uint32_t)h2_pick((int)len, k0, hi);
            const uint32_t offk = (uint32_t)h2_pick((int)off, k0, hi);
            const double bxk = VALUES ? __hiloint2double(h2_pick(bxhi, k0, hi), h2_pick(bxlo, k0, hi)) : 0.0;
            P.act[i] = k0 + (hi ? 1 : 0) < nseg && (uint32_t)gl < lenk;
            const int32_t q = P.act[i] ? abk + gl : 0;   // clamped: nnz(A) > 0
            P.row[i] = (abl & 4) ? (uint32_t)q * 2654435761u >> 12 : (uint32_t)Ai[q];
            P.v[i] = (abl & 4) ? bxk : (VALUES ? bxk * Ax[q] : 0.0);
            P.t[i] = offk + (uint32_t)gl;
        }
    }
}

__device__ __forceinline__ uint32_t h2_exclusive(uint32_t v, int lane) {   // exclusive wave scan
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    return inc - v;
}

// Persistent workgroups, columns ci = blockIdx.x, + gridDim.x, ...  Everything a column needs from memory is requested
// ahead of its turn and rides along with the work on earlier columns: descriptors three columns ahead, B entries two,
// the extents of the A columns they name between one and two, and the products themselves (rows and values of A, the
// only loads that depend on data) ONE column ahead -- issued before the current column's inserts, used after its
// copy-out.  (Gathered just in time they cost 4.6 of 11 ms.)  (Round 4: the three barriers of the column loop as LDS-only
// barriers -- __syncthreads() also waits for the global loads in flight, i.e. for the next column's products -- 9.8 ms either
// way: the prefetch is not what the loop waits for.)
template <bool VALUES>
__global__ __launch_bounds__(256) void k_sg_hash2(int slots_, const int4 *__restrict__ info, int32_t ncols,
                                                  const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                  const double *__restrict__ Ax, const int32_t *__restrict__ Bi,
                                                  const double *__restrict__ Bx, int32_t *__restrict__ count,
                                                  int32_t *__restrict__ tmp_i, double *__restrict__ tmp_x, int abl) {
    // abl (timing experiments, ablation build only; results are wrong unless 0): 1 = no inserts, 2 = no stores to the
    // product-order buffer, 4 = rows made up instead of gathered, 8 = no first-touch ranking
#ifndef CSX_ABLATION
    abl = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t slots = (uint32_t)slots_;
    double *val = reinterpret_cast<double *>(smem);
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + (VALUES ? (size_t)slots * 8 : 0));
    uint32_t *tmin = keys + slots;
    uint32_t *ecnt = tmin + slots;              // first touches per entry of B(:,j): H2_MAXSEG words
    const int tid = threadIdx.x, lane = tid & 63, gl = tid & 31;
    const bool hi = (lane & 32) != 0;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t below = (1u << gl) - 1u;     // lanes of the half-wave before this one
    const int32_t G = (int32_t)gridDim.x;
    const int4 none = make_int4(-1, 0, 0, 0);
    int32_t ci = blockIdx.x;                    // the host launches at most ncols workgroups
    int4 cur = info[ci];
    int4 nxt = ci + G < ncols ? info[ci + G] : none;
    int4 nn = ci + 2 * G < ncols ? info[ci + 2 * G] : none;
    // entry `lane` of a column, in every wave: current column (products P0), next column (ab1 / len1 / bx1)
    H2Prod P0;
    int nseg0 = cur.z - cur.y;
    {
        int32_t ab = 0;
        uint32_t len = 0;
        double bx = 0.0;
        if (lane < nseg0) {
            const int32_t c = Bi[cur.y + lane];
            ab = Ap[c];
            len = (uint32_t)(Ap[c + 1] - ab);
            if (VALUES) bx = Bx[cur.y + lane];
        }
        h2_fetch<VALUES>(P0, ab, len, h2_exclusive(len, lane), bx, nseg0, wu, hi, gl, Ai, Ax, abl);
    }
    int nseg1 = nxt.x >= 0 ? nxt.z - nxt.y : 0;
    int32_t ab1 = 0;
    uint32_t len1 = 0;
    double bx1 = 0.0;
    if (lane < nseg1) {
        const int32_t c = Bi[nxt.y + lane];
        ab1 = Ap[c];
        len1 = (uint32_t)(Ap[c + 1] - ab1);
        if (VALUES) bx1 = Bx[nxt.y + lane];
    }
    for (uint32_t k = tid; k < slots; k += 256) {
        keys[k] = SG_UNSET;
        tmin[k] = SG_UNSET;
        if (VALUES) val[k] = 0.0;
    }
    __syncthreads();
    for (;;) {
        // ---- requests for later columns ----
        const int4 n3 = ci + 3 * G < ncols ? info[ci + 3 * G] : none;
        const int nseg2 = nn.x >= 0 ? nn.z - nn.y : 0;
        int32_t c2 = 0;
        double bx2 = 0.0;
        if (lane < nseg2) {
            c2 = Bi[nn.y + lane];
            if (VALUES) bx2 = Bx[nn.y + lane];
        }
        H2Prod P1;                              // the next column's products: on their way while this column is done
        h2_fetch<VALUES>(P1, ab1, len1, h2_exclusive(len1, lane), bx1, nseg1, wu, hi, gl, Ai, Ax, abl);
        // ---- this column: insert ----
        const int32_t j = cur.x;
        const int nseg = nseg0;
        uint32_t ps[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ps[i] = 0;
        if (abl & 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) P0.act[i] = P0.act[i] && P0.row[i] == 0x12345u;
        }
        h2_insert<VALUES, 4>(keys, tmin, val, slots, P0.row, P0.t, P0.v, P0.act, ps);
        if (nseg > 32) h2_insert<VALUES, 2>(keys, tmin, val, slots, P0.row + 4, P0.t + 4, P0.v + 4, P0.act + 4, ps + 4);
        if (nseg > 48) h2_insert<VALUES, 2>(keys, tmin, val, slots, P0.row + 6, P0.t + 6, P0.v + 6, P0.act + 6, ps + 6);
        __syncthreads();
        // ---- first touches: a product is one iff it holds its row's smallest product number; per entry of B their
        //      count (one ballot), inside an entry the rank is the number of first touches in the lanes before ----
        bool first[8];
        uint32_t rank_in[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            first[i] = false;
            rank_in[i] = 0;
            if (8 * i < nseg && !(abl & 8)) {   // uniform
                first[i] = P0.act[i] && tmin[ps[i]] == P0.t[i];
                const unsigned long long bal = __ballot(first[i]);
                const uint32_t m32 = hi ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                rank_in[i] = (uint32_t)__popc(m32 & below);
                if (gl == 0) ecnt[8 * i + 2 * wu + (hi ? 1 : 0)] = (uint32_t)__popc(m32);
            }
        }
        __syncthreads();
        // ---- position in the column = first touches of the entries before + rank inside the entry; emit; wipe ----
        const uint32_t ec = (lane < ((nseg + 7) & ~7) && !(abl & 8)) ? ecnt[lane] : 0u;
        const uint32_t eex = h2_exclusive(ec, lane);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)(eex + ec), 63);
        const int64_t base = cur.w;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (8 * i < nseg) {   // uniform
                const uint32_t pre = (uint32_t)h2_pick((int)eex, 8 * i + 2 * wu, hi);
                if (first[i]) {
                    const uint32_t pos = pre + rank_in[i];
                    const uint32_t sl = ps[i];
                    if (!(abl & 2)) tmp_i[base + pos] = (int32_t)P0.row[i];
                    keys[sl] = SG_UNSET;
                    tmin[sl] = SG_UNSET;
                    if (VALUES) {
                        if (!(abl & 2)) tmp_x[base + pos] = val[sl];
                        val[sl] = 0.0;
                    }
                }
            }
        }
        if (tid == 0) count[j] = (abl & 10) ? 0 : (int32_t)cnt;
        if (abl & 8)
            for (uint32_t k = tid; k < slots; k += 256) {
                keys[k] = SG_UNSET;
                tmin[k] = SG_UNSET;
                if (VALUES) val[k] = 0.0;
            }
        // ---- extents of the A columns named two columns ahead (their B entries have arrived by now) ----
        int32_t ab2 = 0, e2 = 0;
        if (lane < nseg2) {
            ab2 = Ap[c2];
            e2 = Ap[c2 + 1];
        }
        __syncthreads();                        // table clean, ecnt read: the next column may start
        ci += G;
        if (ci >= ncols) break;
        cur = nxt;
        nxt = nn;
        nn = n3;
        P0 = P1;
        nseg0 = nseg1;
        nseg1 = nseg2;
        ab1 = ab2;
        len1 = (uint32_t)(e2 - ab2);
        bx1 = bx2;
    }
}

template <bool VALUES>
static int launch_hash2(int slots, const Csc *A, const Csc *B, const int4 *info, int32_t ncols, int32_t *count,
                        int32_t *tmp_i, double *tmp_x, int cap_per_cu = 8) {
    if (ncols <= 0) return CSX_OK;
    const size_t lds = (size_t)(slots + (slots & 1)) * (VALUES ? 16 : 8) + H2_MAXSEG * 4 + 64;
    auto kern = k_sg_hash2<VALUES>;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(cap_per_cu, (160 * 1024) / (int64_t)(lds + 256)));
    const int64_t grid = std::min<int64_t>(ncols, (int64_t)ctx().cus * per_cu);
    const int abl = ablation_env("CSX_SG_ABL") ? std::atoi(ablation_env("CSX_SG_ABL")) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx().stream, slots, info, ncols, A->p, A->i, A->x,
                       B->i, B->x, count, tmp_i, tmp_x, abl);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

// one wave per column: move its rows (and sums) from the product-order buffer to their place in C.  A column's length is
// count[j] when `count` is given (the chunked path: the NEXT chunk's scan is rewriting Cp[j + 1] of a chunk's last column
// while that chunk is compacted on the second stream), else Cp[j + 1] - Cp[j].
__global__ __launch_bounds__(256) void k_sg_compact(const uint32_t *__restrict__ cols, int32_t ncols,
                                                    const int32_t *__restrict__ toff, const int32_t *__restrict__ Cp,
                                                    const int32_t *__restrict__ count,
                                                    const int32_t *__restrict__ tmp_i, const double *__restrict__ tmp_x,
                                                    int32_t *__restrict__ Ci, double *__restrict__ Cx) {
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= ncols) return;
    const int32_t j = (int32_t)cols[w];
    const int64_t src = toff[j], dst = Cp[j];
    const int32_t cnt = count ? count[j] : Cp[j + 1] - Cp[j];
    int32_t k = lane;
    for (; k + 192 < cnt; k += 256) {   // four loads of each array in flight per lane
        int32_t ri[4];
        double rx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ri[u] = tmp_i[src + k + 64 * u];
        if (tmp_x) {
#pragma unroll
            for (int u = 0; u < 4; u++) rx[u] = tmp_x[src + k + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) Ci[dst + k + 64 * u] = ri[u];
        if (tmp_x) {
#pragma unroll
            for (int u = 0; u < 4; u++) Cx[dst + k + 64 * u] = rx[u];
        }
    }
    for (; k < cnt; k += 64) {
        Ci[dst + k] = tmp_i[src + k];
        if (tmp_x) Cx[dst + k] = tmp_x[src + k];
    }
}

static size_t h1_lds_bytes(int slots, bool values) {   // slots = 3/2 * (products allowed in the bin)
    return (size_t)(slots + (slots & 1)) * (values ? 16 : 8) + H1_SEG * (8 + 4 + 4 + 4) + (H1_MAXP / 32) * 8 + 64 +
           (size_t)(slots * 2 / 3) * 2 + 16;
}

template <bool VALUES>
static int launch_hash1(int slots, const Csc *A, const Csc *B, const int4 *info, int32_t ncols, int32_t *count,
                        int32_t *tmp_i, double *tmp_x, int cap_per_cu = 8) {
    if (ncols <= 0) return CSX_OK;
    const size_t lds = h1_lds_bytes(slots, VALUES);
    auto kern = k_sg_hash1<VALUES>;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    // exactly the workgroups that are resident at once (LDS-limited, at most 8 x 4 waves per CU)
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(cap_per_cu, (160 * 1024) / (int64_t)(lds + 256)));
    const int64_t grid = std::min<int64_t>(ncols, (int64_t)ctx().cus * per_cu);
    const int abl = ablation_env("CSX_SG_ABL") ? std::atoi(ablation_env("CSX_SG_ABL")) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx().stream, slots, info, ncols, A->p, A->i, A->x,
                       B->i, B->x, count, tmp_i, tmp_x, abl);
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
    for (int b = 0; b < SG_NBINS; b++) {
        const int32_t nb = bin_ptr[b + 1] - bin_ptr[b];
        const bool hashed = b >= SG_BIN_HASH0 && b < SG_BIN_NARROW0 + SG_NARROW_BINS;
        const int hb = b >= SG_BIN_NARROW0 ? b - SG_BIN_NARROW0 : b - SG_BIN_HASH0;
        if (b == SG_BIN_EMPTY || (hashed && skip_hash)) continue;  // hash bins: done by the one-pass kernel
        if (!hashed && b != SG_BIN_DENSE && b != SG_BIN_GLOBAL) continue;
        const int kind = b == SG_BIN_DENSE ? ACC_LDS_DENSE : (b == SG_BIN_GLOBAL ? ACC_GLOBAL_DENSE : ACC_LDS_HASH);
        int slots = 0;  // two-pass hash tables: power of two, load <= 1/2
        if (hashed)
            for (slots = 1024; slots < 2 * sg_hash_limit(hb); slots <<= 1) {}
        CSX_TRY((launch_bin<NUMERIC, VALUES>(kind, slots, A, B, cols + bin_ptr[b], nb, count, Cp, Ci, Cx, g_tmin, g_val)));
    }
    return CSX_OK;
}

// ---- chunked one-pass path (opt-in, spgemm.chunks >= 2) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sg_chunk_key(int32_t n, int32_t chunk_cols, const uint32_t *__restrict__ bin,
                                                      uint32_t *__restrict__ key) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) key[j] = (uint32_t)(j / chunk_cols) * (uint32_t)SG_NBINS + bin[j];
}

// Cp[c0 .. c1] (a chunk's exclusive scan, starting at 0) += base[cur]; base[cur ^ 1] = base[cur] + the chunk's total
__global__ __launch_bounds__(256) void k_sg_add_base(int32_t *__restrict__ Cp, int32_t c0, int32_t c1,
                                                     unsigned long long *base, int cur) {
    const int64_t j = c0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > c1) return;
    const unsigned long long b = base[cur], v = b + (unsigned long long)Cp[j];
    Cp[j] = (int32_t)(v > 0x7FFFFFFFull ? 0x7FFFFFFFull : v);
    if (j == c1) base[cur ^ 1] = v;
}

// The chunked form of the one-pass path: columns in ascending chunks; while chunk c + 1 is hashed (an instruction- and
// LDS-bound kernel that leaves the memory system mostly idle) chunk c -- its counts scanned behind chunk c - 1's end, which
// IS C.p for those columns -- is compacted into C.i / C.x on a second stream (a pure copy).  C.i / C.x are allocated for
// the upper bound "one entry per product" (the reference grows C the same way, csparse.py:1630-1631, and trims at the
// end).  MEASURED SLOWER than the unchunked path on S (1M x 1M, 32 per column: 14.9 ms unchunked; 15.1 / 15.4 / 16.2 ms
// with 2 / 4 / 8 chunks; with the hash kernels held to three workgroups per CU so that the copy's waves find registers,
// 16.1 - 16.6 ms): the hash kernel fills the vector registers of every SIMD (122 VGPRs x 4 waves), so the copy's waves
// only get in at a chunk's tail, and every chunk adds the ramp and tail of five persistent launches (one per bin).  Kept
// as an option for matrices whose columns sit in one bin; off by default (profiles/r03_ablation.md, section 2).
// Every non-empty column is in a hash bin (the caller checks).  bin / colid / hprod / count as multiply_device made them.
static hipStream_t g_sg_stream = nullptr;
#ifndef SG_OVERLAP_CAP
#define SG_OVERLAP_CAP 4
#endif

static int multiply_chunked(const Csc *A, const Csc *B, Csc *C, bool values, const uint32_t *bin, const uint32_t *colid,
                            const int32_t *hprod, unsigned long long P, int32_t *count) {
    hipStream_t s = ctx().stream;
    const int32_t n = B->n;
    const size_t per = values ? 12 : 4;
    int64_t nchunks = std::max(2, std::min(ctx().opt.spgemm_chunks, 64));   // opt-in (spgemm.chunks >= 2)
    nchunks = std::min<int64_t>(nchunks, std::max<int64_t>(2, n / 2048));
    const int32_t chunk_cols = (int32_t)(((int64_t)n + nchunks - 1) / nchunks);
    nchunks = ((int64_t)n + chunk_cols - 1) / chunk_cols;
    if (!g_sg_stream) CSX_HIP(hipStreamCreateWithFlags(&g_sg_stream, hipStreamNonBlocking));
    DevScope tmp;
    int32_t *toff = nullptr, *ptr_d = nullptr, *tmp_i = nullptr;
    uint32_t *key = nullptr, *skey = nullptr, *scol = nullptr;
    int4 *info = nullptr;
    double *tmp_x = nullptr;
    unsigned long long *base = nullptr;
    CSX_TRY(tmp.alloc(&toff, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&key, (size_t)n));
    CSX_TRY(tmp.alloc(&skey, (size_t)n));
    CSX_TRY(tmp.alloc(&scol, (size_t)n));
    CSX_TRY(tmp.alloc(&info, (size_t)n));
    CSX_TRY(tmp.alloc(&base, 2));
    const int32_t nkeys = (int32_t)nchunks * SG_NBINS;
    CSX_TRY(tmp.alloc(&ptr_d, (size_t)nkeys + 1));
    int64_t tot = 0;
    CSX_TRY(scan_exclusive_i32(hprod, toff, n, &tot));
    hipLaunchKernelGGL(k_sg_chunk_key, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, n, chunk_cols, bin, key);
    CSX_TRY(stable_sort_by_key(key, colid, nullptr, n, (uint32_t)nkeys, skey, scol, nullptr));
    CSX_TRY(boundaries_from_sorted(skey, n, nkeys, ptr_d));
    std::vector<int32_t> ptr((size_t)nkeys + 1);
    CSX_HIP(hipMemcpyAsync(ptr.data(), ptr_d, ptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipMemsetAsync(base, 0, 2 * sizeof(unsigned long long), s));
    CSX_TRY(tmp.alloc(&tmp_i, (size_t)P));
    if (values) CSX_TRY(tmp.alloc(&tmp_x, (size_t)P));
    hipLaunchKernelGGL(k_sg_colinfo, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, scol, n, B->p, toff, info);
    // C.i / C.x for the upper bound (one entry per product); C.p is written chunk by chunk
    int32_t *Ci = nullptr;
    double *Cx = nullptr;
    CSX_TRY(dalloc(&Ci, (size_t)P));
    C->i = Ci;
    if (values) {
        CSX_TRY(dalloc(&Cx, (size_t)P));
        C->x = Cx;
    }
    CSX_HIP(hipStreamSynchronize(s));
    std::vector<hipEvent_t> ev((size_t)nchunks, nullptr);
    hipEvent_t done = nullptr;
    int st = CSX_OK;
    for (auto &e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) st = CSX_ERUNTIME;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) st = CSX_ERUNTIME;
    for (int64_t c = 0; c < nchunks && st == CSX_OK; c++) {
        const int32_t *pc = ptr.data() + c * SG_NBINS;
        for (int hb = 0; hb < SG_HASH_BINS && st == CSX_OK; hb++) {
            const int32_t lo = pc[SG_BIN_HASH0 + hb], nb = pc[SG_BIN_HASH0 + hb + 1] - lo;
            const int slots = sg_hash_limit(hb) * 3 / 2;
            st = values ? launch_hash1<true>(slots, A, B, info + lo, nb, count, tmp_i, tmp_x, SG_OVERLAP_CAP)
                        : launch_hash1<false>(slots, A, B, info + lo, nb, count, tmp_i, nullptr, SG_OVERLAP_CAP);
        }
        for (int hb = 0; hb < SG_NARROW_BINS && st == CSX_OK; hb++) {
            const int32_t lo = pc[SG_BIN_NARROW0 + hb], nb = pc[SG_BIN_NARROW0 + hb + 1] - lo;
            const int slots = sg_hash_limit(hb) * 3 / 2;
            st = values ? launch_hash2<true>(slots, A, B, info + lo, nb, count, tmp_i, tmp_x, SG_OVERLAP_CAP)
                        : launch_hash2<false>(slots, A, B, info + lo, nb, count, tmp_i, nullptr, SG_OVERLAP_CAP);
        }
        if (st != CSX_OK) break;
        const int32_t c0 = (int32_t)(c * chunk_cols), c1 = (int32_t)std::min<int64_t>(n, (c + 1) * (int64_t)chunk_cols);
        st = scan_exclusive_i32(count + c0, C->p + c0, c1 - c0, nullptr);
        if (st != CSX_OK) break;
        hipLaunchKernelGGL(k_sg_add_base, dim3((unsigned)((c1 - c0 + 256) / 256)), dim3(256), 0, s, C->p, c0, c1, base, (int)(c & 1));
        if (hipEventRecord(ev[(size_t)c], s) != hipSuccess || hipStreamWaitEvent(g_sg_stream, ev[(size_t)c], 0) != hipSuccess) {
            st = CSX_ERUNTIME;
            break;
        }
        const int32_t lo = pc[SG_BIN_HASH0], nh = pc[SG_BIN_NARROW0 + SG_NARROW_BINS] - lo;
        if (nh > 0)
            hipLaunchKernelGGL(k_sg_compact, dim3((unsigned)(((int64_t)nh + 3) / 4)), dim3(256), 0, g_sg_stream, scol + lo, nh, toff,
                               C->p, count, tmp_i, tmp_x, Ci, Cx);
    }
    // the context's stream continues behind the last copy (and nothing is freed before it has finished)
    if (hipEventRecord(done, g_sg_stream) != hipSuccess || hipStreamWaitEvent(s, done, 0) != hipSuccess) st = CSX_ERUNTIME;
    unsigned long long total = 0;
    if (hipMemcpyAsync(&total, base + (nchunks & 1), sizeof total, hipMemcpyDeviceToHost, s) != hipSuccess) st = CSX_ERUNTIME;
    if (hipStreamSynchronize(g_sg_stream) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    if (done) (void)hipEventDestroy(done);
    if (st == CSX_OK && hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
    CSX_TRY(st);
    if (total > 0x7FFFFFFFull) {
        set_error("cs_multiply: the product has %llu entries (int32 indices)", total);
        return CSX_EINVAL;
    }
    C->nnz = (int32_t)total;
    // Much of the capacity unused (many products per entry): move to arrays of the exact size, as cs_sprealloc(C, 0) does
    if ((P - total) * per > ((size_t)256 << 20) && P > total + total / 4) {
        int32_t *Ci2 = nullptr;
        double *Cx2 = nullptr;
        CSX_TRY(dalloc(&Ci2, (size_t)total));
        CSX_HIP(hipMemcpyAsync(Ci2, Ci, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (values) {
            CSX_TRY(dalloc(&Cx2, (size_t)total));
            CSX_HIP(hipMemcpyAsync(Cx2, Cx, (size_t)total * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        CSX_HIP(hipStreamSynchronize(s));
        dfree(Ci);
        dfree(Cx);
        C->i = Ci2;
        C->x = Cx2;
    }
    return CSX_OK;
}

// ---- opt-in: C.x in the reference's summation order (spgemm.ordered) ---------------------------------------------------
// cs_scatter adds the products of an entry in the order it meets them (csparse.py:1979-1988): B(:,j) in storage order,
// inside it A(:,k) in storage order; the first product is ASSIGNED (x[i] = beta * Ax[p]), the later ones added, each
// product and each sum rounded on its own.  The hash kernels add with LDS atomics in arrival order, so their x differs
// from the reference's in the last bits and from run to run.  This pass recomputes x from the finished pattern in
// exactly the reference's order: one wave per column of C walks the products in order, one entry of B(:,j) at a time
// (its A column's entries one per lane); the products of one such batch go to distinct rows unless A(:,k) holds a row
// twice, and then the lanes of equal rows take their turns in lane order.  A row's position in C(:,j) comes from a hash
// map in LDS (columns of at most SGO_MAX entries, sums in LDS too) or from a dense map in memory (longer columns, sums
// straight into C.x).  Bit-identical to the unmodified reference (tests/test_gpu_multiply.py); several times slower than the atomics.
constexpr int SGO_MAX = 2048, SGO_SLOTS = 4096;

__device__ __forceinline__ int sgo_rank_of_equal_rows(int32_t row, bool active, int lane, int *rounds) {
    // number of earlier active lanes holding the same row; *rounds = 1 + the largest such number in the wave
    int before = 0;
    for (int u = 0; u < 63; u++) {
        const int32_t ru = __builtin_amdgcn_readlane(row, u);
        const bool au = ((__ballot(active) >> u) & 1ull) != 0;
        if (au && active && lane > u && ru == row) before++;
    }
    int mx = before;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    *rounds = mx + 1;
    return before;
}

template <bool BIG>
__global__ __launch_bounds__(64) void k_sg_values_ordered(int32_t n, int32_t m, const int32_t *__restrict__ Ap,
                                                          const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                          const int32_t *__restrict__ Bp, const int32_t *__restrict__ Bi,
                                                          const double *__restrict__ Bx, const int32_t *__restrict__ Cp,
                                                          const int32_t *__restrict__ Ci, double *__restrict__ Cx,
                                                          int32_t *__restrict__ gmap) {
#pragma clang fp contract(off)
    __shared__ int32_t hkey[BIG ? 1 : SGO_SLOTS];
    __shared__ int32_t hpos[BIG ? 1 : SGO_SLOTS];
    __shared__ double acc[BIG ? 1 : SGO_MAX];
    __shared__ unsigned long long seen[BIG ? 1 : SGO_MAX / 64];
    const int lane = threadIdx.x;
    int32_t *map = BIG ? gmap + (int64_t)blockIdx.x * m : nullptr;
    for (int32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const int32_t c0 = Cp[j], cnt = Cp[j + 1] - c0;
        if (cnt == 0 || (cnt > SGO_MAX) != BIG) continue;
        // the map row -> position in C(:,j)
        if (!BIG) {
            for (int s = lane; s < SGO_SLOTS; s += 64) hkey[s] = -1;
            for (int s = lane; s < SGO_MAX / 64; s += 64) seen[s] = 0ull;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            for (int32_t q = lane; q < cnt; q += 64) {
                const int32_t row = Ci[c0 + q];
                uint32_t s = (uint32_t)(row * 0x9E3779B1u) >> 20;      // 12 bits
                for (;;) {
                    const int32_t prev = atomicCAS(&hkey[s], -1, row);
                    if (prev == -1) break;
                    s = (s + 1) & (SGO_SLOTS - 1);
                }
                hpos[s] = q;
            }
        } else {
            // dense map in memory: row -> cnt + position until the position is first touched, then the position
            for (int32_t q = lane; q < cnt; q += 64) {
                map[Ci[c0 + q]] = cnt + q;
                Cx[c0 + q] = 0.0;
            }
            __threadfence_block();
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int32_t pb = Bp[j]; pb < Bp[j + 1]; pb++) {           // B(:,j) in storage order
            const int32_t k = Bi[pb];
            const double beta = Bx[pb];
            for (int32_t a0 = Ap[k]; a0 < Ap[k + 1]; a0 += 64) {   // A(:,k) in storage order, 64 entries at a time
                const int32_t q = a0 + lane;
                const bool active = q < Ap[k + 1];
                const int32_t row = active ? Ai[q] : -1;
                const double prod = active ? beta * Ax[q] : 0.0;
                int rounds = 1;
                const int before = sgo_rank_of_equal_rows(row, active, lane, &rounds);
                int32_t pos = 0;
                if (active && !BIG) {
                    uint32_t s = (uint32_t)(row * 0x9E3779B1u) >> 20;
                    while (hkey[s] != row) s = (s + 1) & (SGO_SLOTS - 1);
                    pos = hpos[s];
                }
                for (int r = 0; r < rounds; r++) {
                    if (active && before == r) {
                        if (!BIG) {
                            const unsigned long long bit = 1ull << (pos & 63);
                            if (seen[pos >> 6] & bit) acc[pos] = acc[pos] + prod;
                            else {
                                acc[pos] = prod;                   // the first product is assigned (csparse.py:1986)
                                atomicOr(&seen[pos >> 6], bit);
                            }
                        } else {
                            const int32_t v = map[row];
                            if (v >= cnt) {                        // first touch of this position
                                Cx[c0 + v - cnt] = prod;
                                map[row] = v - cnt;
                            } else {
                                Cx[c0 + v] = Cx[c0 + v] + prod;
                            }
                        }
                    }
                    if (BIG) __threadfence_block();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!BIG) {
            for (int32_t q = lane; q < cnt; q += 64) Cx[c0 + q] = acc[q];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the same for columns of at most 2 048 entries, round 4 -------------------------------------------------------------
// The kernel above takes 250 ms on S (1M columns of ~990 entries): 48 KB of LDS per wave leave three waves to a CU, every
// batch of products pays its own memory round trip, and the test for equal rows inside a batch -- 63 readlanes -- runs for
// every batch although a column of A almost never holds a row twice.  Here: (1) whether a column of A holds a row twice
// inside one of its batches of 64 is found ONCE per column of A (k_sg_dup_cols); batches of clean columns update their 64
// sums at once; (2) the table is sized to the column (512 / 2 560 / 4 096 slots for up to 256 / 1 280 / 2 048 entries: 7 / 34 /
// 54 KB per wave), sums are kept per SLOT, so no position array; (3) a wave keeps its column of B in registers (entry l in
// lane l), requests the A columns of eight entries together and looks up their slots side by side before the updates,
// which go in order.  S: 250 -> 42 ms for the pass (cs_multiply with "spgemm.ordered": 265 -> 57 ms).  Same operation order per entry of C, so still bit-identical to the reference.
__global__ __launch_bounds__(256) void k_sg_dup_cols(int32_t nA, int nbits, const int32_t *__restrict__ Ap,
                                                     const int32_t *__restrict__ Ai, uint8_t *__restrict__ dup) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= nA) return;
    bool any = false;
    for (int32_t a0 = Ap[k]; a0 < Ap[k + 1]; a0 += 64) {
        const bool active = a0 + lane < Ap[k + 1];
        const uint32_t row = active ? (uint32_t)Ai[a0 + lane] : 0u;
        unsigned long long peers = __ballot(active);
        for (int b = 0; b < nbits; b++) {
            const unsigned long long bal = __ballot((row >> b) & 1u);
            peers &= ((row >> b) & 1u) ? bal : ~bal;
        }
        if (active && __popcll(peers) > 1) any = true;
    }
    if (lane == 0) dup[k] = 0;
    if (__ballot(any) != 0ull && lane == 0) dup[k] = 1;
}

// slot of a row in a table of SLOTS entries (any size: the high half of a 32 x 32 bit product)
template <int SLOTS>
__device__ __forceinline__ uint32_t sgo_slot(int32_t row) {
    return __umulhi((uint32_t)row * 0x9E3779B1u, (uint32_t)SLOTS);
}

template <int SLOTS, int MAXCNT>
__global__ __launch_bounds__(64) void k_sg_values_ordered2(int32_t n, int32_t lo_cnt, const int32_t *__restrict__ Ap,
                                                           const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                           const uint8_t *__restrict__ dupA, const int32_t *__restrict__ Bp,
                                                           const int32_t *__restrict__ Bi, const double *__restrict__ Bx,
                                                           const int32_t *__restrict__ Cp, const int32_t *__restrict__ Ci,
                                                           double *__restrict__ Cx) {
#pragma clang fp contract(off)
    constexpr int G = 8;   // entries of B whose A columns are requested together
    __shared__ int32_t hkey[SLOTS];
    __shared__ double acc[SLOTS];
    __shared__ uint16_t slotq[MAXCNT];
    __shared__ unsigned long long seen[(SLOTS + 63) / 64];
    const int lane = threadIdx.x;
    for (int32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const int32_t c0 = Cp[j], cnt = Cp[j + 1] - c0;
        if (cnt <= lo_cnt || cnt > MAXCNT) continue;
        for (int sl = lane; sl < SLOTS; sl += 64) hkey[sl] = -1;
        for (int sl = lane; sl < (SLOTS + 63) / 64; sl += 64) seen[sl] = 0ull;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int32_t q = lane; q < cnt; q += 64) {
            const int32_t row = Ci[c0 + q];
            uint32_t sl = sgo_slot<SLOTS>(row);
            for (;;) {
                const int32_t prev = atomicCAS(&hkey[sl], -1, row);
                if (prev == -1) break;
                sl = sl + 1 == SLOTS ? 0 : sl + 1;
            }
            slotq[q] = (uint16_t)sl;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int32_t b0 = Bp[j], b1 = Bp[j + 1];
        for (int32_t pb0 = b0; pb0 < b1; pb0 += 64) {                 // B(:,j) in storage order, 64 entries to a round
            const int nbe = min(64, b1 - pb0);
            int32_t ka = 0, kb = 0, kd = 0;
            double beta_l = 0.0;
            if (lane < nbe) {
                const int32_t k = Bi[pb0 + lane];
                beta_l = Bx[pb0 + lane];
                ka = Ap[k];
                kb = Ap[k + 1];
                kd = dupA[k];
            }
            for (int e0 = 0; e0 < nbe; e0 += G) {
                int32_t rows[G];
                double vals[G];
#pragma unroll
                for (int u = 0; u < G; u++) {                         // the first batch of each of the next G columns of A
                    const int ee = min(e0 + u, nbe - 1);
                    const int32_t a0 = __builtin_amdgcn_readlane(ka, ee), a1 = __builtin_amdgcn_readlane(kb, ee);
                    const bool in = e0 + u < nbe && a0 + lane < a1;
                    rows[u] = in ? Ai[a0 + lane] : -1;
                    vals[u] = in ? Ax[a0 + lane] : 0.0;
                }
                // the slots of all G batches first (independent probe chains side by side), then the updates in order
                uint32_t sls[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    sls[u] = 0;
                    if (rows[u] >= 0) {
                        uint32_t sl = sgo_slot<SLOTS>(rows[u]);
                        while (hkey[sl] != rows[u]) sl = sl + 1 == SLOTS ? 0 : sl + 1;
                        sls[u] = sl;
                    }
                }
#pragma unroll
                for (int u = 0; u < G; u++) {
                    if (e0 + u >= nbe) break;
                    const int32_t a0 = __builtin_amdgcn_readlane(ka, e0 + u), a1 = __builtin_amdgcn_readlane(kb, e0 + u);
                    const bool dupk = __builtin_amdgcn_readlane(kd, e0 + u) != 0;
                    const double beta = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(beta_l), e0 + u),
                                                         __builtin_amdgcn_readlane(__double2loint(beta_l), e0 + u));
                    for (int32_t a = a0; a < a1; a += 64) {           // A(:,k) in storage order, 64 entries at a time
                        int32_t row = rows[u];
                        double val = vals[u];
                        uint32_t sl = sls[u];
                        if (a != a0) {                                // a long column of A: its later batches come as they are needed
                            const bool in = a + lane < a1;
                            row = in ? Ai[a + lane] : -1;
                            val = in ? Ax[a + lane] : 0.0;
                            if (in) {
                                sl = sgo_slot<SLOTS>(row);
                                while (hkey[sl] != row) sl = sl + 1 == SLOTS ? 0 : sl + 1;
                            }
                        }
                        const bool active = row >= 0;
                        const double prod = beta * val;
                        int rounds = 1, before = 0;
                        if (dupk) before = sgo_rank_of_equal_rows(row, active, lane, &rounds);
                        for (int r = 0; r < rounds; r++) {
                            if (active && before == r) {
                                const unsigned long long bit = 1ull << (sl & 63);
                                const unsigned long long old = atomicOr(&seen[sl >> 6], bit);
                                acc[sl] = (old & bit) ? acc[sl] + prod : prod;   // the first product is assigned (csparse.py:1986)
                            }
                            if (rounds > 1) {
                                __builtin_amdgcn_s_waitcnt(0xc07f);
                                __builtin_amdgcn_wave_barrier();
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int32_t q = lane; q < cnt; q += 64) Cx[c0 + q] = acc[slotq[q]];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// flag[c] |= some column of C falls into class c: 0: 1 .. 256 entries, 1: .. 1 280, 2: .. 2 048, 3: more
__global__ void k_sg_count_classes(int32_t n, const int32_t *__restrict__ Cp, int *flag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t cnt = Cp[j + 1] - Cp[j];
    if (cnt == 0) return;
    const int c = cnt <= 256 ? 0 : (cnt <= 1280 ? 1 : (cnt <= SGO_MAX ? 2 : 3));
    if (!flag[c]) flag[c] = 1;
}

static int values_in_reference_order(const Csc *A, const Csc *B, Csc *C) {
    if (!C->x || C->nnz == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int32_t n = C->n, m = C->m;
    DevScope tmp;
    int *flag = nullptr, h[4] = {0, 0, 0, 0};
    uint8_t *dup = nullptr;
    CSX_TRY(tmp.alloc(&flag, 4));
    CSX_TRY(tmp.alloc(&dup, (size_t)A->n + 1));
    CSX_HIP(hipMemsetAsync(flag, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(k_sg_count_classes, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, n, C->p, flag);
    CSX_HIP(hipMemcpyAsync(h, flag, sizeof h, hipMemcpyDeviceToHost, s));
    int nbits = 1;
    while (nbits < 31 && (1ll << nbits) < (long long)m) nbits++;
    hipLaunchKernelGGL(k_sg_dup_cols, dim3((unsigned)(((int64_t)A->n + 3) / 4)), dim3(256), 0, s, A->n, nbits, A->p, A->i, dup);
    CSX_HIP(hipStreamSynchronize(s));
    const int cus = ctx().cus;
#define CSX_ORD(SLOTS, MAXCNT, LO, WAVES)                                                                                      \
    hipLaunchKernelGGL((k_sg_values_ordered2<SLOTS, MAXCNT>), dim3((unsigned)std::min<int64_t>(n, (int64_t)cus * (WAVES))), dim3(64), \
                       0, s, n, LO, A->p, A->i, A->x, dup, B->p, B->i, B->x, C->p, C->i, C->x)
    // (tables at most half full: with linear probing a wave waits for its LONGEST probe chain, and at two thirds full --
    // 1 536 slots for 1 024 entries, seven waves to a CU instead of five -- that made the pass six times slower, not faster)
    if (h[0]) CSX_ORD(512, 256, 0, 20);          //  7 KB of LDS per wave
    if (h[1]) CSX_ORD(2560, 1280, 256, 4);       // 34 KB: four waves to a CU (S: 1 024 +- 100 entries per column -- with the
                                                 // class cut at 1 024 a quarter of S went to the 54 KB class: 32 of 49 ms)
    if (h[2]) CSX_ORD(4096, 2048, 1280, 2);      // 54 KB
#undef CSX_ORD
    if (h[3]) {
        const int64_t waves = std::min<int64_t>(n, 128);
        int32_t *gmap = nullptr;
        CSX_TRY(tmp.alloc(&gmap, (size_t)waves * (size_t)m));
        hipLaunchKernelGGL(k_sg_values_ordered<true>, dim3((unsigned)waves), dim3(64), 0, s, n, m, A->p, A->i, A->x, B->p, B->i, B->x,
                           C->p, C->i, C->x, gmap);
    }
    CSX_HIP(hipStreamSynchronize(s));
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

int multiply_device(const Csc *A, const Csc *B, Csc *C) {
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
    int4 *info = nullptr;
    double *g_val = nullptr, *tmp_x = nullptr;
    unsigned long long *too_big = nullptr;  // [0] columns with >= 2^32 products, [1] products in hash-bin columns
    int st = dalloc(&bin, (size_t)n);
    if (st == CSX_OK) st = dalloc(&colid, (size_t)n);
    if (st == CSX_OK) st = dalloc(&sbin, (size_t)n);
    if (st == CSX_OK) st = dalloc(&scol, (size_t)n);
    if (st == CSX_OK) st = dalloc(&bin_ptr_d, SG_NBINS + 1);
    if (st == CSX_OK) st = dalloc(&count, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&hprod, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&too_big, 2);
    int32_t bin_ptr[SG_NBINS + 1] = {0};
    unsigned long long big[2] = {0, 0};
    if (st == CSX_OK) {
        (void)hipMemsetAsync(too_big, 0, 2 * sizeof(unsigned long long), s);
        (void)hipMemsetAsync(count, 0, ((size_t)n + 1) * sizeof(int32_t), s);
        hipLaunchKernelGGL(k_sg_products, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, A->p, B->p, B->i, m,
                           bin, colid, hprod, too_big);
        hipLaunchKernelGGL(k_sum_i32, dim3(512), dim3(256), 0, s, hprod, (int64_t)n, too_big + 1);
        st = stable_sort_by_key(bin, colid, nullptr, n, SG_NBINS, sbin, scol, nullptr);
    }
    if (st == CSX_OK) st = boundaries_from_sorted(sbin, n, SG_NBINS, bin_ptr_d);
    if (st == CSX_OK) {
        if (hipMemcpyAsync(bin_ptr, bin_ptr_d, (SG_NBINS + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(big, too_big, sizeof big, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    if (st == CSX_OK && big[0]) st = CSX_EINVAL;  // a column with >= 2^32 products
    // one-pass path for the hash bins when its product-order buffer (12 B per product) is affordable
    const int32_t hash_lo = bin_ptr[SG_BIN_HASH0], nhash = bin_ptr[SG_BIN_NARROW0 + SG_NARROW_BINS] - hash_lo;
    bool onepass = false;
    if (st == CSX_OK && nhash > 0 && big[1] < 0x7FFFFFF0ull && ctx().opt.spgemm_one_pass) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)big[1] * (values ? 12 : 4);
        size_t idle_b = 0;
        pool_stats(&idle_b, nullptr);   // idle blocks of the caching allocator are reusable
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < (free_b + idle_b) / 3) onepass = true;
    }
    // all non-empty columns hashed, and a product-order buffer far bigger than the Infinity Cache: take the columns in chunks
    const bool chunked = st == CSX_OK && onepass && ctx().opt.spgemm_chunks > 1 &&
                         bin_ptr[SG_BIN_DENSE + 1] == bin_ptr[SG_BIN_DENSE] && bin_ptr[SG_BIN_GLOBAL + 1] == bin_ptr[SG_BIN_GLOBAL] &&
                         n >= 4096 &&
                         big[1] * (values ? 24 : 8) < (size_t)0x7FFFFFF0ull * 24;
    if (chunked) {
        st = multiply_chunked(A, B, C, values, bin, colid, hprod, big[1], count);
        if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
        for (void *q : {(void *)hprod, (void *)bin, (void *)colid, (void *)sbin, (void *)scol, (void *)bin_ptr_d, (void *)count,
                        (void *)too_big})
            dfree(q);
        return st;
    }
    if (st == CSX_OK && onepass) {
        st = dalloc(&toff, (size_t)n + 1);
        int64_t tot = 0;
        if (st == CSX_OK) st = scan_exclusive_i32(hprod, toff, n, &tot);
        if (st == CSX_OK) st = dalloc(&tmp_i, (size_t)big[1]);
        if (st == CSX_OK && values) st = dalloc(&tmp_x, (size_t)big[1]);
        if (st == CSX_OK) st = dalloc(&info, (size_t)nhash);
        if (st == CSX_OK)
            hipLaunchKernelGGL(k_sg_colinfo, dim3((unsigned)((nhash + 255) / 256)), dim3(256), 0, s, scol + hash_lo, nhash,
                               B->p, toff, info);
        for (int hb = 0; hb < SG_HASH_BINS && st == CSX_OK; hb++) {
            const int32_t lo = bin_ptr[SG_BIN_HASH0 + hb], nb = bin_ptr[SG_BIN_HASH0 + hb + 1] - lo;
            const int slots = sg_hash_limit(hb) * 3 / 2;   // load factor <= 2/3
            st = values ? launch_hash1<true>(slots, A, B, info + (lo - hash_lo), nb, count, tmp_i, tmp_x)
                        : launch_hash1<false>(slots, A, B, info + (lo - hash_lo), nb, count, tmp_i, nullptr);
        }
        for (int hb = 0; hb < SG_NARROW_BINS && st == CSX_OK; hb++) {
            const int32_t lo = bin_ptr[SG_BIN_NARROW0 + hb], nb = bin_ptr[SG_BIN_NARROW0 + hb + 1] - lo;
            const int slots = sg_hash_limit(hb) * 3 / 2;
            st = values ? launch_hash2<true>(slots, A, B, info + (lo - hash_lo), nb, count, tmp_i, tmp_x)
                        : launch_hash2<false>(slots, A, B, info + (lo - hash_lo), nb, count, tmp_i, nullptr);
        }
    }
    const int32_t nglobal = bin_ptr[SG_BIN_GLOBAL + 1] - bin_ptr[SG_BIN_GLOBAL];
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
        hipLaunchKernelGGL(k_sg_compact, dim3((unsigned)(((int64_t)nhash + 3) / 4)), dim3(256), 0, s, scol + hash_lo,
                           nhash, toff, C->p, nullptr, tmp_i, tmp_x, C->i, C->x);
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
    dfree(info);
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
    if (st == CSX_OK && ctx().opt.spgemm_ordered) st = values_in_reference_order(A, B, C);
    if (st != CSX_OK) {
        free_csc(C);
        return st;
    }
    *out = put(K_CSC, C);
    return CSX_OK;
}
