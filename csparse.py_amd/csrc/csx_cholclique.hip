// Forests of cliques: cs_schol + cs_chol (csparse.py:2051-2072, 561-619) for matrices whose elimination forest is a set of
// CLIQUES ON CONSECUTIVE COLUMNS -- block-diagonal SPD matrices with dense blocks, batches of small dense systems (G-spd,
// BASELINE config 5).  For such a matrix nothing of the general pattern machine (csx_cholsym.hip: postorder, start pairs,
// two radix sorts, row-subtree walks, two more sorts) is needed:
//
//   u[k] = the smallest row of column k's upper part (rows <= k; k itself when there is none).
//   The forest is a set of cliques on consecutive columns  <=>  u[0] = 0 and, for k > 0, u[k] = k (k starts a block)
//   or u[k] = u[k-1] (k continues the block of k-1).
//   (=>: column a of L receives no fill, so it is full iff every later column of the block holds A(a, k).  <=: every column
//   of the block reaches the block's first column, no column reaches further back, so etree = a chain per block and every
//   column of L is full below the diagonal inside its block.)
//
// Then parent[k] = k + 1 inside a block, -1 at its end; column k of L holds rows k .. end of the block (cs_chol appends
// rows in ascending order, the diagonal first: L.i follows from cp alone); and the values are one read of A's upper part
// and one write of L: k_chol_clique keeps a whole block (<= 64 columns) in the registers of one wave.
#include <chrono>
#include <cstdio>

#include "csx_internal.h"
#include "csx_cholclique.h"

namespace csx {

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// ---- recognition ---------------------------------------------------------------------------------------------------
// one wave per column: u[k]; flags[0] |= the upper part of some column is not strictly ascending (duplicates, unsorted,
// a negative row) -- the block kernel scatters a column's entries in parallel and needs them distinct
__global__ __launch_bounds__(256) void k_clique_min(int32_t n, const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                    int32_t *__restrict__ u, int *flags) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= n) return;
    const int32_t b = Ap[k], e = Ap[k + 1];
    int32_t mn = (int32_t)k, last = -1;
    bool bad = false;
    for (int32_t p0 = b; p0 < e; p0 += 64) {
        const int32_t p = p0 + lane;
        int32_t i = 0x7fffffff;
        if (p < e) i = Ai[p];
        const bool up = i <= (int32_t)k;
        const unsigned long long bal = __ballot(up);
        if (bal == 0ull) continue;
        const unsigned long long below = bal & ((1ull << lane) - 1ull);
        const int prevlane = below ? 63 - __clzll((long long)below) : 0;
        int32_t prev = __shfl(i, prevlane);
        if (!below) prev = last;
        if (up && i <= prev) bad = true;
        last = __shfl(i, 63 - __clzll((long long)bal));
        if (up) mn = min(mn, i);
    }
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
    if (__ballot(bad) != 0ull && lane == 0) flags[0] = 1;
    if (lane == 0) u[k] = mn;
}

// thread per column: the rule above (flags[1] |= broken), block starts, parent, the last column of every block
__global__ __launch_bounds__(256) void k_clique_mark(int32_t n, const int32_t *__restrict__ u, int32_t *__restrict__ is_start,
                                                     int32_t *__restrict__ parent, int32_t *__restrict__ end_of, int *flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t uk = u[k];
    const bool ok = k == 0 ? uk == 0 : (uk == (int32_t)k || uk == u[k - 1]);
    if (!ok || uk < 0) {
        flags[1] = 1;
        is_start[k] = 0;
        parent[k] = -1;
        return;
    }
    is_start[k] = uk == (int32_t)k ? 1 : 0;
    const bool last = k == n - 1 || u[k + 1] == (int32_t)k + 1;
    parent[k] = last ? -1 : (int32_t)k + 1;
    if (last) end_of[uk] = (int32_t)k;
}

// thread per column: column counts of L; block list; size statistics (stats[0] = widest block, lnz in *lnz)
__global__ __launch_bounds__(256) void k_clique_counts(int32_t n, const int32_t *__restrict__ u, const int32_t *__restrict__ is_start,
                                                       const int32_t *__restrict__ block_id, const int32_t *__restrict__ end_of,
                                                       int32_t *__restrict__ count, int32_t *__restrict__ start, int *stats,
                                                       unsigned long long *lnz) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t a = u[k], e = end_of[a];
    count[k] = e - (int32_t)k + 1;
    if (is_start[k]) {
        const int32_t bs = e - a + 1;
        start[block_id[k]] = (int32_t)k;
        atomicMax(&stats[0], bs);
        atomicAdd(lnz, (unsigned long long)bs * (unsigned long long)(bs + 1) / 2ull);
    }
    if (k == n - 1) start[block_id[k] + is_start[k]] = n;
}

void free_clique(CliqueForest *F) {
    dfree(F->parent);
    dfree(F->cp);
    dfree(F->start);
    F->parent = F->cp = F->start = nullptr;
}

int clique_forest(const Csc *A, CliqueForest *F, bool *ok) {
    *ok = false;
    const int32_t n = A->n;
    if (n <= 0 || A->m != A->n) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *u = nullptr, *is_start = nullptr, *block_id = nullptr, *end_of = nullptr, *count = nullptr;
    int *flags = nullptr;   // [0] not ascending, [1] not a clique forest, [2] widest block; [4..5] lnz (64 bits)
    CSX_TRY(tmp.alloc(&u, (size_t)n));
    CSX_TRY(tmp.alloc(&is_start, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&block_id, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&end_of, (size_t)n));
    CSX_TRY(tmp.alloc(&count, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&flags, 8));
    CSX_TRY(dalloc(&F->parent, (size_t)n));
    CSX_HIP(hipMemsetAsync(flags, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(k_clique_min, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, A->p, A->i, u, flags);
    hipLaunchKernelGGL(k_clique_mark, dim3(blocks_for(n)), dim3(256), 0, s, n, u, is_start, F->parent, end_of, flags);
    int h[8] = {0};
    CSX_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (h[1]) {
        free_clique(F);
        return CSX_OK;
    }
    int64_t nblocks = 0;
    CSX_TRY(scan_exclusive_i32(is_start, block_id, n, &nblocks));
    CSX_TRY(dalloc(&F->start, (size_t)nblocks + 1));
    CSX_TRY(dalloc(&F->cp, (size_t)n + 1));
    hipLaunchKernelGGL(k_clique_counts, dim3(blocks_for(n)), dim3(256), 0, s, n, u, is_start, block_id, end_of, count,
                       F->start, flags + 2, (unsigned long long *)(flags + 4));
    CSX_TRY(scan_exclusive_i32(count, F->cp, n, nullptr));
    CSX_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    unsigned long long lnz = 0;
    std::memcpy(&lnz, h + 4, sizeof lnz);
    if (lnz > 0x7fffffffull) {   // L does not fit int32 indices: the general path reports it
        free_clique(F);
        return CSX_OK;
    }
    F->n = n;
    F->nblocks = (int32_t)nblocks;
    F->max_bs = h[2];
    F->lnz = (int64_t)lnz;
    F->ascending = h[0] == 0;
    *ok = true;
    return CSX_OK;
}

__global__ __launch_bounds__(256) void k_clique_compare(int32_t n, const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                        const int32_t *__restrict__ ca, const int32_t *__restrict__ cb, int *bad) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n) return;
    if (ca[q] != cb[q] || (q < n && pa[q] != pb[q])) *bad = 1;
}

// cs_chol(A, S): the caller's S.parent / S.cp (host arrays) must be this forest's; uploaded and compared on the device
int clique_matches_host(const CliqueForest &F, const int32_t *parent, const int32_t *cp, bool *same) {
    *same = false;
    hipStream_t s = ctx().stream;
    const int32_t n = F.n;
    DevScope tmp;
    int32_t *dp = nullptr, *dc = nullptr;
    int *bad = nullptr;
    CSX_TRY(tmp.alloc(&dp, (size_t)n));
    CSX_TRY(tmp.alloc(&dc, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&bad, 1));
    CSX_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
    CSX_HIP(hipMemcpyAsync(dp, parent, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(dc, cp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_clique_compare, dim3(blocks_for((int64_t)n + 1)), dim3(256), 0, s, n, dp, F.parent, dc, F.cp, bad);
    int h = 0;
    CSX_HIP(hipMemcpyAsync(&h, bad, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    *same = h == 0;
    return CSX_OK;
}

// ---- values --------------------------------------------------------------------------------------------------------
// One wave per block of bs <= 64 columns; lane r owns ROW r of the block, a[c] = element (r, c) of the lower triangle.
//
// Loading: cs_chol reads the UPPER part of C (csparse.py:593-595): entry C(i, k), i <= k, is element (k, i) of the lower
// triangle, i.e. lane k's register i -- a coalesced read of column k of A delivers one lane's registers spread over the
// wave, so 16 columns at a time pass through an LDS tile (column-major with an odd stride) and the 16 lanes that own them
// read their rows back.  Entries the pattern lacks are zeros (fill).
//
// Factoring: right-looking, in panels of 8 columns.  Every element receives its updates - L(r, j) L(c, j) for j = 0, 1, ...
// in ascending order, multiply then subtract, then one division by the pivot: the operation sequence of cs_chol's up-looking
// row solve on a chain (csparse.py:598-612; cs_ereach hands the columns over in ascending order), so L.x is bit-identical to
// it.  L(c, j) comes from lane c by v_readlane and enters the multiplication as a scalar operand.  The register file has no
// dynamic index, and the whole triangle unrolled would be 90 KB of code against 64 KB of instruction cache: the loop body is
// written for a WINDOW whose first eight registers are the current panel -- factor those, update the rest of the window by
// them (groups of eight columns, skipped when they lie beyond the block), store the panel, slide the window by eight
// registers -- 18 KB of code, executed bs / 8 times, no update outside the triangle's rectangle of live groups.
//
// Storing: column g of the block goes to L.x / L.i at Lp[c0] + g bs - g (g - 1) / 2, rows ascending: runs of up to 512 / 256 bytes.
constexpr int CQ_WAVES = 4;
constexpr int CQ_CH = 16;     // columns staged at a time
constexpr int CQ_LD = 65;     // doubles per staged column

__device__ __forceinline__ double cq_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64 * CQ_WAVES, 3) void k_chol_clique(const int32_t *__restrict__ start, int32_t nblocks, int32_t n,
                                                                 const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                                 const double *__restrict__ Ax, const int32_t *__restrict__ Lp,
                                                                 int32_t *__restrict__ Li, double *__restrict__ Lx, int *notspd) {
#pragma clang fp contract(off)
    __shared__ double s_tile[CQ_WAVES][CQ_CH * CQ_LD];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t = (int64_t)blockIdx.x * CQ_WAVES + w;
    if (t >= nblocks) return;   // no workgroup barrier below
    const int32_t c0 = __builtin_amdgcn_readfirstlane(start[t]);
    const int32_t bs = __builtin_amdgcn_readfirstlane(start[t + 1]) - c0;
    double *tile = s_tile[w];
    double a[64];
#pragma unroll
    for (int c = 0; c < 64; c++) a[c] = 0.0;
    // ---- load ----
#pragma unroll 1
    for (int q0 = 0; q0 < bs; q0 += CQ_CH) {
        for (int e = lane; e < CQ_CH * CQ_LD; e += 64) tile[e] = 0.0;
        const int32_t colp = Ap[min(c0 + q0 + lane, n)];   // lanes 0 .. 16 matter
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int kk0 = 0; kk0 < CQ_CH; kk0 += 4) {
            int32_t ii[4], pb[4], pe[4];
            double vv[4];
#pragma unroll
            for (int uu = 0; uu < 4; uu++) {
                pb[uu] = __builtin_amdgcn_readlane(colp, kk0 + uu);
                pe[uu] = __builtin_amdgcn_readlane(colp, kk0 + uu + 1);
                if (q0 + kk0 + uu >= bs) pe[uu] = pb[uu];
                const int32_t p = pb[uu] + lane;
                ii[uu] = p < pe[uu] ? Ai[p] : 0x7fffffff;
            }
#pragma unroll
            for (int uu = 0; uu < 4; uu++) {
                const int32_t col = c0 + q0 + kk0 + uu;
                const bool up = ii[uu] <= col && ii[uu] >= c0;
                vv[uu] = up ? Ax[pb[uu] + lane] : 0.0;
                ii[uu] = up ? ii[uu] - c0 : -1;
            }
#pragma unroll
            for (int uu = 0; uu < 4; uu++)
                if (ii[uu] >= 0) tile[(kk0 + uu) * CQ_LD + ii[uu]] = vv[uu];
#pragma unroll
            for (int uu = 0; uu < 4; uu++) {   // columns of more than 64 entries (a lower part with duplicates, say)
                const int32_t col = c0 + q0 + kk0 + uu;
                for (int32_t p0 = pb[uu] + 64; p0 < pe[uu]; p0 += 64) {
                    const int32_t p = p0 + lane;
                    const int32_t i = p < pe[uu] ? Ai[p] : 0x7fffffff;
                    if (i <= col && i >= c0) tile[(kk0 + uu) * CQ_LD + i - c0] = Ax[p];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const bool mine = (lane >> 4) == (q0 >> 4);
        const double *row = tile + (lane & 15) * CQ_LD;
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const double v = row[c];
            if (mine) a[c] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- factor ----
    const int64_t base = Lp[c0];
#pragma unroll 1
    for (int J = 0; J < bs; J += 8) {
#pragma unroll
        for (int jw = 0; jw < 8; jw++) {
            const int g = J + jw;
            if (g < bs) {
                const double d = cq_bcast(a[jw], g);
                if (d <= 0.0 && lane == 0) atomicMin(notspd, c0 + g);   // csparse.py:612
                const double ljj = sqrt(d);
                const double l = a[jw] / ljj;
                a[jw] = lane == g ? ljj : l;
#pragma unroll
                for (int cw = jw + 1; cw < 8; cw++) {
                    const double pr = a[jw] * cq_bcast(a[jw], J + cw);
                    a[cw] = a[cw] - pr;
                }
            }
        }
#pragma unroll
        for (int gq = 1; gq < 8; gq++) {
            if (J + 8 * gq < bs) {
#pragma unroll
                for (int cc = 0; cc < 8; cc++) {
                    const int cw = 8 * gq + cc;
#pragma unroll
                    for (int jw = 0; jw < 8; jw++) {
                        const double pr = a[jw] * cq_bcast(a[jw], J + cw);
                        a[cw] = a[cw] - pr;
                    }
                }
            }
        }
#pragma unroll
        for (int jw = 0; jw < 8; jw++) {
            const int g = J + jw;
            if (g < bs && lane >= g && lane < bs) {
                const int64_t q = base + (int64_t)g * bs - (int64_t)g * (g - 1) / 2 + (lane - g);
                Lx[q] = a[jw];
                Li[q] = c0 + lane;
            }
        }
#pragma unroll
        for (int c = 0; c < 56; c++) a[c] = a[c + 8];
    }
}

int chol_clique_numeric(const Csc *A, const CliqueForest &F, Csc *L, int *d_notspd) {
    hipStream_t s = ctx().stream;
    if (F.nblocks == 0) return CSX_OK;
    hipLaunchKernelGGL(k_chol_clique, dim3((unsigned)((F.nblocks + CQ_WAVES - 1) / CQ_WAVES)), dim3(64 * CQ_WAVES), 0, s, F.start,
                       F.nblocks, A->n, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

// ---- is this L the factor of a forest of equal dense blocks?  (cholsol plan) ------------------------------------------
// one wave per column: rows j, j + 1, ... contiguous; the count falls by one from column to column inside a block;
// stats[0] |= no, stats[1] = max over block starts of the count, stats[2] = min
__global__ __launch_bounds__(256) void k_clique_factor_shape(int32_t n, const int32_t *__restrict__ Lp,
                                                             const int32_t *__restrict__ Li, int *stats) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t b = Lp[j], cnt = Lp[j + 1] - b;
    bool bad = cnt < 1 || cnt > n - (int32_t)j;
    if (!bad) {
        for (int32_t q = lane; q < cnt; q += 64)
            if (Li[b + q] != (int32_t)j + q) bad = true;
        if (lane == 0) {
            const bool first = j == 0 || Lp[j] - Lp[j - 1] == 1;     // the column before ended its block
            if (!first && Lp[j] - Lp[j - 1] != cnt + 1) bad = true;
            if (first) {
                atomicMax(&stats[1], cnt);
                atomicMin(&stats[2], cnt);
            }
        }
    }
    if (bad) stats[0] = 1;
}

int clique_factor_block_size(const Csc *L, int32_t *bs) {
    *bs = 0;
    const int32_t n = L->n;
    if (n <= 0 || L->m != n || !L->x) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int *stats = nullptr;
    CSX_TRY(tmp.alloc(&stats, 4));
    int h[4] = {0, 0, 0x7fffffff, 0};
    CSX_HIP(hipMemcpyAsync(stats, h, sizeof h, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_clique_factor_shape, dim3(blocks_for((int64_t)n * 64)), dim3(256), 0, s, n, L->p, L->i, stats);
    CSX_HIP(hipMemcpyAsync(h, stats, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (h[0] == 0 && h[1] == h[2] && h[1] >= 1 && n % h[1] == 0) *bs = h[1];
    return CSX_OK;
}

}  // namespace csx
